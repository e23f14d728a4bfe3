#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
date
tools/profile_bench.sh r3_c2
date
tools/profile_bench.sh r3_c4 --context-views 3 --batch 4
date
tools/profile_bench.sh r3_c5 --size 512 --batch 2
date
