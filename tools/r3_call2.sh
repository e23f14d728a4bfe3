#!/bin/bash
# round-3 GPU call 2: the fast-form tile kernels (default build) through the raster / decoder / graph
# tests, then the variants (tests + interleaved A/B), then the side-stream weight-gradient A/B
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
L=$PWD/pixelsplat_amd
RT="tests/test_raster_gpu.py tests/test_raster_configs_gpu.py tests/test_decoder_gpu.py tests/test_graph_gpu.py"
echo "== default: raster tests"; date
timeout 900 python -m pytest $RT -m gpu -q --timeout 600 2>&1 | tail -15
for tag in q2w6 q1w6 w4; do
  echo "-- $tag"
  PIXELSPLAT_HIP_LIB=$L/libps_$tag.so timeout 400 python -m pytest tests/test_raster_gpu.py tests/test_raster_configs_gpu.py::test_config1_256 tests/test_raster_configs_gpu.py::test_config0_64 tests/test_decoder_gpu.py -m gpu -x -q 2>&1 | tail -3
done
echo "== A/B tiles"; date
tools/ab_env.sh r3b_ab "" "PIXELSPLAT_HIP_LIB=$L/libps_st.so" "PIXELSPLAT_HIP_LIB=$L/libps_nofast.so" "PIXELSPLAT_HIP_LIB=$L/libps_w4.so" "PIXELSPLAT_HIP_LIB=$L/libps_q2w6.so" "PIXELSPLAT_HIP_LIB=$L/libps_q2w5.so" "PIXELSPLAT_HIP_LIB=$L/libps_q1w6.so" "PS_WGRAD_SIDE=0" 2>&1 | sed -e "s/PIXELSPLAT_HIP_LIB=[^ ]*libps_//" | cut -c1-420
echo "== epipolar tests (side-stream weight gradients)"; date
timeout 900 python -m pytest tests/test_epipolar_gpu.py tests/test_head_gpu.py tests/test_depth_gpu.py -m gpu -q --timeout 600 2>&1 | tail -5
date
