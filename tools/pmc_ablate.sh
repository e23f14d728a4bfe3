#!/bin/bash
# SQ counters of kernels of VARIANT libraries (tools/build_variant.sh <tag> -D...), on the GPU box -- e.g. the timing
# ablations of rounds 2 / 3 (PS_ABLATE*: parts of a kernel's work compiled out; those macros left the source in round 4,
# the numbers are in profiles/r2f_tiles_ablation.txt, r2f_attention_ablation.txt, r3_attention_counters.txt).
# usage: tools/pmc_ablate.sh [kernel-name substring, default "tiles_"] [variant tags, default "abl2 abl1"]
#        e.g. tools/pmc_ablate.sh epipolar_attn "att3 att2 att1"
# Prints per kernel the counters per launch in millions (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are
# quad-cycles).  Counters in their own pass, kernel trace only (no sys / hip tracing beside --pmc).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
pat=${1:-tiles_}; tags=${2:-"abl2 abl1"}
cd /tmp && export TMPDIR=/tmp
for v in $tags; do
  rm -rf /tmp/pmc_$v
  PIXELSPLAT_HIP_LIB=$R/pixelsplat_amd/libps_$v.so timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmc_$v -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probes --launch eager > /dev/null 2>&1
  python - "$v" "$pat" <<'P'
import csv, glob, sys, collections
v, pat = sys.argv[1], sys.argv[2]
f = glob.glob(f'/tmp/pmc_{v}/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row['Kernel_Name']
    if pat not in k:
        continue
    acc[k[:48]][row['Counter_Name']] += float(row['Counter_Value'])
    if row['Counter_Name'] == 'SQ_INSTS_VALU':
        n[k[:48]] += 1
for k in acc:
    print(v, k, {c: round(x / n[k] / 1e6, 1) for c, x in acc[k].items()}, 'launches', n[k])
P
done
