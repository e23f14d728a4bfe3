#!/bin/bash
# rocprofv3 kernel durations of the gemm_tn variants; summary into gpurun_out/<tag>_v<variant>.csv
tag=${1:-prof_gemm_tn}
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in ${VARIANTS:-2 3}; do
  rm -rf /tmp/prof_$v
  PYTHONPATH=. PS_GEMM_TN_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o out --output-format csv -- python tools/ab_gemm_tn.py > /tmp/prof_$v.log 2>&1
  f=$(find /tmp/prof_$v -name '*kernel_stats.csv' | head -1)
  head -8 "$f" | cut -c1-200 > gpurun_out/${tag}_v$v.csv
  echo "== variant $v"; cat gpurun_out/${tag}_v$v.csv
  # per-launch durations of the partial kernel, grouped by grid size
  t=$(find /tmp/prof_$v -name '*kernel_trace.csv' | head -1)
  python - "$t" <<'PY' | tee -a gpurun_out/${tag}_v$v.csv
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemm_tn' in r['Kernel_Name']:
        d[(r['Kernel_Name'][:40], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', ''))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(d.items()):
    v.sort()
    print(k, 'n=%d median=%.1f us min=%.1f' % (len(v), v[len(v)//2] / 1e3, v[0] / 1e3))
PY
done
