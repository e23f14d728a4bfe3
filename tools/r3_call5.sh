#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
L=$PWD/pixelsplat_amd
RT="tests/test_raster_gpu.py tests/test_raster_configs_gpu.py tests/test_decoder_gpu.py tests/test_graph_gpu.py"
echo "== default: raster tests"; date
timeout 900 python -m pytest $RT -m gpu -q --timeout 600 2>&1 | tail -8
for tag in q2w5; do
  echo "-- $tag"
  PIXELSPLAT_HIP_LIB=$L/libps_$tag.so timeout 400 python -m pytest tests/test_raster_gpu.py tests/test_raster_configs_gpu.py::test_config1_256 tests/test_raster_configs_gpu.py::test_config0_64 tests/test_decoder_gpu.py -m gpu -x -q 2>&1 | tail -3
done
echo "== A/B tiles"; date
tools/ab_env.sh r3e_ab "" "PIXELSPLAT_HIP_LIB=$L/libps_st.so" "PIXELSPLAT_HIP_LIB=$L/libps_nofast.so" "PIXELSPLAT_HIP_LIB=$L/libps_q2w5.so" "PIXELSPLAT_HIP_LIB=$L/libps_q2nf.so" 2>&1 | sed -e "s/PIXELSPLAT_HIP_LIB=[^ ]*libps_//" | cut -c1-300
echo "== counters"
cp $L/libpixelsplat_hip.so $L/libps_def.so
tools/pmc_ablate.sh tiles_forward "def nofast" 2>&1 | tee gpurun_out/r3e_pmc_tiles.txt
date
