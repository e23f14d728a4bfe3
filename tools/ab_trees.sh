#!/bin/bash
# Interleaved A/B of bench.py between this tree ("base") and a second checkout of the repository ("alt", default
# ./_old_tree with its own built library) on the GPU box -- for changes that touch the host side or the state layout,
# where tools/ab_env.sh's library swap is not enough.   tools/ab_trees.sh <reps> [alt tree] [bench args]
cd "$(dirname "$0")/.."
reps=${1:-2}; alt=${2:-_old_tree}
if [ $# -ge 2 ]; then shift 2; else shift $#; fi
mkdir -p gpurun_out
for r in $(seq 1 $reps); do
  for var in base alt; do
    if [ $var = alt ]; then dir=$alt; else dir=.; fi
    (cd $dir && timeout -k 10 300 python bench.py --no-cpu-baseline --no-probes "$@") > gpurun_out/ab_env_$var.json 2> gpurun_out/ab_env_$var.err
    python - $var $r <<'PY'
import json, sys
var, r = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(f"gpurun_out/ab_env_{var}.json").read().strip().splitlines()[-1])
    k = d["kernels_ms"]
    print(f"rep {r} {var:>4}: step {d['ms_per_step']:.3f}  (B) {d['paths'].get('raster_only_ms_per_step')}  (A) {d['paths'].get('epipolar_only_ms_per_step')}  "
          + "  ".join(f"{g} {k[g]}" for g in ("preprocess_forward", "depth_sort", "tile_bins", "tiles_forward", "tiles_backward", "preprocess_backward") if g in k)
          + f"  check {d.get('step_check', {}).get('ok')}", flush=True)
except Exception as e:
    print(f"rep {r} {var}: FAILED {e!r}", flush=True)
    print(open(f"gpurun_out/ab_env_{var}.err").read()[-1500:], flush=True)
PY
  done
done
