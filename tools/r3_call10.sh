#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
L=$PWD/pixelsplat_amd
tools/ab_env.sh r3j_ab "" "PIXELSPLAT_HIP_LIB=$L/libps_tg8.so" "PIXELSPLAT_HIP_LIB=$L/libps_tg2.so" 2>&1 | sed -e "s/PIXELSPLAT_HIP_LIB=[^ ]*libps_//" | cut -c1-120
tools/kstats.sh r3j 2>&1 | grep -i "epipolar\|bin_\|tile_order\|total" | head -20
