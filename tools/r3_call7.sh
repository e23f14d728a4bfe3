#!/bin/bash
# round-3 GPU call 7: attention kernels -- counters of the ablated builds (instruction shares of the
# phases), a kernel-level profile of the step (A) alone incl. every torch kernel,
# and the roctx ranges as rocprofv3 --marker-trace sees them
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== attention counters"; date
tools/pmc_ablate.sh epipolar_attn "def att1 att2 att3" 2>&1 | tee gpurun_out/r3g_pmc_attn.txt
echo "== kernel stats of the eager step"; date
tools/kstats.sh r3g 2>&1 | head -70 | tee gpurun_out/r3g_kstats.txt
echo "== roctx"; date
PS_ROCTX=1 timeout 200 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d gpurun_out/r3g_roctx -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-probes --launch eager > gpurun_out/r3g_roctx.log 2>&1
ls gpurun_out/r3g_roctx/* | head; for f in gpurun_out/r3g_roctx/*marker*stats*.csv gpurun_out/r3g_roctx/*marker_api_trace.csv; do [ -f "$f" ] && { echo "-- $f"; head -30 "$f"; }; done
python - <<'PY'
import csv, glob, collections
fs = glob.glob('gpurun_out/r3g_roctx/**/*marker_api_trace.csv', recursive=True)
if fs:
    c = collections.Counter(); t = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        n = r.get('Function') or r.get('Name') or ''
        c[n] += 1
        try: t[n] += (int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        except Exception: pass
    with open('gpurun_out/r3g_roctx_ranges.txt', 'w') as f:
        for n, k in c.most_common():
            line = f'{n:40s} ranges {k:5d}  host time inside {t[n]/1e6:9.3f} ms'
            print(line); f.write(line + '\n')
PY
rm -rf gpurun_out/r3g_roctx
date
