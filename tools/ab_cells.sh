#!/bin/bash
# A/B of the tile forward on the GPU box: the 8x8-quadrant forward (PS_FORWARD_QUADRANTS=1) against the 4x4-cell
# forward's build variants (PS_CELLS_VARIANT, csrc/raster_cells.hip), interleaved `reps` times; one line per run.
# usage: tools/ab_cells.sh [reps, default 2] [extra bench.py arguments]
cd "$(dirname "$0")/.."
reps=${1:-2}; shift
mkdir -p gpurun_out
for r in $(seq 1 $reps); do
  for var in old 0 2; do
    if [ $var = old ]; then export PS_FORWARD_QUADRANTS=1; unset PS_CELLS_VARIANT; else unset PS_FORWARD_QUADRANTS; export PS_CELLS_VARIANT=$var; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-probes "$@" > gpurun_out/ab_cells_$var.json 2> gpurun_out/ab_cells_$var.err
    python - $var $r <<'PY'
import json, sys
var, r = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(f"gpurun_out/ab_cells_{var}.json").read().strip().splitlines()[-1])
    k = d["kernels_ms"]
    print(f"rep {r} variant {var:>3}: step {d['ms_per_step']:.3f} ms  (B) {d['paths'].get('raster_only_ms_per_step')}  "
          f"tiles_forward {k.get('tiles_forward')}  tiles_backward {k.get('tiles_backward')}  "
          f"tile_bins {k.get('tile_bins')}  step_check {d.get('step_check', {}).get('ok')}", flush=True)
except Exception as e:
    print(f"rep {r} variant {var}: FAILED {e!r}", flush=True)
    print(open(f"gpurun_out/ab_cells_{var}.err").read()[-1500:], flush=True)
PY
  done
done
