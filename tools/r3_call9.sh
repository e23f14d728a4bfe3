#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
L=$PWD/pixelsplat_amd
echo "== epipolar + graph + bench-ranks tests (binned gather)"; date
timeout 1200 python -m pytest tests/test_epipolar_gpu.py tests/test_graph_gpu.py tests/test_bench_ranks_gpu.py -m gpu -q --timeout 900 2>&1 | tail -15
echo "== A/B"; date
tools/ab_env.sh r3h_ab "" "PIXELSPLAT_HIP_LIB=$L/libps_dfold.so" 2>&1 | sed -e "s/PIXELSPLAT_HIP_LIB=[^ ]*libps_//" | cut -c1-600
BENCH_ARGS="--context-views 3 --batch 4" tools/ab_env.sh r3h_ab_c4 "" "PIXELSPLAT_HIP_LIB=$L/libps_dfold.so" 2>&1 | sed -e "s/PIXELSPLAT_HIP_LIB=[^ ]*libps_//" | cut -c1-600
date
