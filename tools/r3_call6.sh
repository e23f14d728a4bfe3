#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
L=$PWD/pixelsplat_amd
echo "== A/B forward split"; date
tools/ab_env.sh r3f_ab "" "PIXELSPLAT_HIP_LIB=$L/libps_q2w5.so" "PIXELSPLAT_HIP_LIB=$L/libps_q2w6.so" "PIXELSPLAT_HIP_LIB=$L/libps_q1w6.so" "PIXELSPLAT_HIP_LIB=$L/libps_q1w8.so" 2>&1 | sed -e "s/PIXELSPLAT_HIP_LIB=[^ ]*libps_//" | cut -c1-300
for tag in q2w5 q1w6; do
  echo "-- $tag"
  PIXELSPLAT_HIP_LIB=$L/libps_$tag.so timeout 400 python -m pytest tests/test_raster_gpu.py tests/test_raster_configs_gpu.py::test_config1_256 tests/test_raster_configs_gpu.py::test_config0_64 tests/test_decoder_gpu.py -m gpu -x -q 2>&1 | tail -3
done
echo "== full suite, default build"; date
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r3f_tests.log 2>&1; tail -4 gpurun_out/r3f_tests.log
date
