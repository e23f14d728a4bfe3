#!/bin/bash
# round-3 GPU call 1: full GPU suite (new batch-7 parity tests), issue-model microbench, the three
# never-run variants (tests + A/B), dfmap ablations, one full bench line
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
L=$PWD/pixelsplat_amd
echo "== tests"; date
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r3a_tests.log 2>&1
tail -25 gpurun_out/r3a_tests.log
echo "== issue model"; date
timeout 200 tools/issue_model > gpurun_out/r3a_issue_model.txt 2>&1; tail -5 gpurun_out/r3a_issue_model.txt
echo "== variant tests"; date
for tag in st bu4; do
  echo "-- $tag"
  PIXELSPLAT_HIP_LIB=$L/libps_$tag.so timeout 300 python -m pytest tests/test_raster_gpu.py tests/test_raster_configs_gpu.py::test_config1_256 tests/test_decoder_gpu.py -m gpu -x -q 2>&1 | tail -3
done
echo "== A/B"; date
tools/ab_env.sh r3a_ab "" "PIXELSPLAT_HIP_LIB=$L/libps_st.so" "PIXELSPLAT_HIP_LIB=$L/libps_bu4.so" "PIXELSPLAT_HIP_LIB=$L/libps_abd1.so" "PIXELSPLAT_HIP_LIB=$L/libps_abd2.so" 2>&1 | sed -e "s/PIXELSPLAT_HIP_LIB=[^ ]*libps_//" | cut -c1-400
echo "== bench"; date
timeout 600 python bench.py > gpurun_out/r3a_c2_bench.json 2> gpurun_out/r3a_c2_bench.err
tail -3 gpurun_out/r3a_c2_bench.err; head -c 600 gpurun_out/r3a_c2_bench.json; echo; date
