#!/bin/bash
# SQ counter passes (separate rocprofv3 --pmc runs, kernel trace only) of a short eager bench run; prints the per-kernel
# means for the kernels whose name contains one of the given patterns.
# usage: tools/pmc_quick.sh <tag> "<pattern> [<pattern> ...]" [bench.py arguments]
set -u
tag=$1; pats=$2; shift 2
out=gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
pmc="$* --steps 2 --warmup 1 --no-cpu-baseline --no-probes --launch eager"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/${tag}_q$i -o p -- python bench.py $pmc > $out/${tag}_q$i.log 2>&1
done
python tools/pmc_sq_summary.py $out/${tag}_q1/*counter_collection.csv $out/${tag}_q2/*counter_collection.csv > $out/${tag}_pmc_sq.json
python tools/pmc_summary.py $out/${tag}_q3/*counter_collection.csv $out/${tag}_q4/*counter_collection.csv > $out/${tag}_pmc_traffic.json
rm -rf $out/${tag}_q[1-4]
python - "$out/${tag}_pmc_sq.json" "$out/${tag}_pmc_traffic.json" $pats <<'PY'
import json, sys
sq = json.load(open(sys.argv[1]))["kernels"]; tr = json.load(open(sys.argv[2]))["kernels"]
for k, v in sq.items():
    if any(p in k for p in sys.argv[3:]):
        print(k[:60], json.dumps(v), json.dumps(tr.get(k, {})))
PY
