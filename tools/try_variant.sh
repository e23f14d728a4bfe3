#!/bin/bash
# On the GPU box: validate and time variant libraries built beforehand with tools/build_variant.sh
# (pixelsplat_amd/libps_<tag>.so travel with the snapshot).  For every tag: the raster / decoder GPU
# tests through the variant (PIXELSPLAT_HIP_LIB), then an interleaved A/B of the bench step against the
# default library.
# usage: tools/try_variant.sh <tag> [<tag> ...]      [TESTS="tests/test_epipolar_gpu.py" BENCH_ARGS=...]
cd "$(dirname "$0")/.."
L=$PWD/pixelsplat_amd
tests=${TESTS:-"tests/test_raster_gpu.py tests/test_raster_configs_gpu.py tests/test_decoder_gpu.py"}
envs=("")
for tag in "$@"; do
  [ -f $L/libps_$tag.so ] || { echo "missing $L/libps_$tag.so (tools/build_variant.sh $tag ...)"; exit 1; }
  echo "== $tag: $tests"
  PIXELSPLAT_HIP_LIB=$L/libps_$tag.so timeout 200 python -m pytest $tests -m gpu -x -q 2>&1 | tail -3
  envs+=("PIXELSPLAT_HIP_LIB=$L/libps_$tag.so")
done
tools/ab_env.sh try_$1 "${envs[@]}" 2>&1 | sed -e "s/PIXELSPLAT_HIP_LIB=[^ ]*libps_//" | cut -c1-320
