#!/bin/bash
# On the GPU box: tools/profile_bench.sh for the three benchmarked BASELINE configurations (bench line with
# the parity / CPU-baseline blocks, rocprofv3 kernel stats, FETCH / WRITE and SQ counter summaries with
# the build stamp, roctx ranges) -> gpurun_out/<tag>_c{2,4,5}_*; copy what matters to profiles/.
# usage: tools/profile_all_configs.sh [tag, default r3]          (~5 GPU-minutes)
cd "$(dirname "$0")/.."
tag=${1:-r3}
mkdir -p gpurun_out
date
tools/profile_bench.sh ${tag}_c2
date
tools/profile_bench.sh ${tag}_c4 --context-views 3 --batch 4
date
tools/profile_bench.sh ${tag}_c5 --size 512 --batch 2
date
