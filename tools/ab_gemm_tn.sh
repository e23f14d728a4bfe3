#!/bin/bash
# A/B of csrc/gemm_tn.hip variants on the GPU box; writes gpurun_out/<tag>.log
tag=${1:-ab_gemm_tn}
mkdir -p gpurun_out
: > gpurun_out/$tag.log
for v in 0 2 3; do
  PYTHONPATH=. PS_GEMM_TN_VARIANT=$v timeout 300 python tools/ab_gemm_tn.py >> gpurun_out/$tag.log 2>&1
done
cat gpurun_out/$tag.log
