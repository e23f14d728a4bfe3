#!/bin/bash
# Interleaved A/B of bench.py under two environments on the GPU box: tools/ab_env.sh <reps> "<VAR=VALUE ...>" [bench args]
# variant "base" = the environment as is, variant "alt" = with the given assignments exported.  One line per run.
cd "$(dirname "$0")/.."
reps=${1:-2}; alt="$2"; shift 2
mkdir -p gpurun_out
for r in $(seq 1 $reps); do
  for var in base alt; do
    if [ $var = alt ]; then pre="env $alt"; else pre="env"; fi
    timeout -k 10 300 $pre python bench.py --no-cpu-baseline --no-probes "$@" > gpurun_out/ab_env_$var.json 2> gpurun_out/ab_env_$var.err
    python - $var $r <<'PY'
import json, sys
var, r = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(f"gpurun_out/ab_env_{var}.json").read().strip().splitlines()[-1])
    k = d["kernels_ms"]; n = d["kernel_launches_per_step"]
    print(f"rep {r} {var:>4}: step {d['ms_per_step']:.3f}  (B) {d['paths'].get('raster_only_ms_per_step')}  (A) {d['paths'].get('epipolar_only_ms_per_step')}  "
          + "  ".join(f"{g} {k[g]}x{n[g]:g}" for g in ("preprocess_forward", "depth_sort", "tile_bins", "tiles_forward", "tiles_backward", "preprocess_backward") if g in k)
          + f"  check {d.get('step_check', {}).get('ok')}", flush=True)
except Exception as e:
    print(f"rep {r} {var}: FAILED {e!r}", flush=True)
    print(open(f"gpurun_out/ab_env_{var}.err").read()[-1500:], flush=True)
PY
  done
done
