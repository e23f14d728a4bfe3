#!/bin/bash
# Builds a variant of the library with extra preprocessor flags into pixelsplat_amd/libps_<tag>.so
# (objects under /tmp), for A/B runs in one box:  PIXELSPLAT_HIP_LIB=$PWD/pixelsplat_amd/libps_<tag>.so
# usage: tools/build_variant.sh <tag> -DPS_BIN_CHUNK=4096 ...
set -e
tag=$1; shift
cd "$(dirname "$0")/.."
obj=/tmp/ps_variant_$tag; mkdir -p $obj
pids=()
for f in pixelsplat_amd/csrc/*.hip; do
  b=$(basename $f .hip); extra=""
  case $b in raster_preprocess|epipolar_geometry) extra="-ffp-contract=off";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I pixelsplat_amd/csrc -Wall -Wno-unused-function $extra "$@" -c $f -o $obj/$b.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $obj/*.o -o pixelsplat_amd/libps_$tag.so
echo pixelsplat_amd/libps_$tag.so
