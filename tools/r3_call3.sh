#!/bin/bash
# round-3 GPU call 3: why is the forward's short form slower in the kernel than in the microbench?
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
L=$PWD/pixelsplat_amd
tools/ab_env.sh r3c_ab "" "PIXELSPLAT_HIP_LIB=$L/libps_nofast.so" "PIXELSPLAT_HIP_LIB=$L/libps_w3.so" "PIXELSPLAT_HIP_LIB=$L/libps_cfast.so" 2>&1 | sed -e "s/PIXELSPLAT_HIP_LIB=[^ ]*libps_//" | cut -c1-330
echo "== counters"
tools/pmc_ablate.sh tiles_forward "def nofast cfast" 2>&1 | tee gpurun_out/r3c_pmc_fwd.txt
