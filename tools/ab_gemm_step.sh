# in-step A/B of the gemm_tn variants: (A) alone and the gemm_tn kernel time inside the step
for v in ${VARIANTS:-0 2 3}; do
PS_GEMM_TN_VARIANT=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --launch eager 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('[gemm_tn variant $v]', 'step', d['ms_per_step'], 'A', d['paths']['epipolar_only_ms_per_step'], 'gemm_tn', k['gemm_tn_splitk'])"
done
