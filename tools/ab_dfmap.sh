python -m pytest tests/test_epipolar_gpu.py -m gpu -x -q 2>&1 | tail -3
for tp in 0 1; do
for cfg in "" "--context-views 3 --batch 4" "--size 512 --batch 2"; do
PS_DFMAP_TWO_PASS=$tp python bench.py $cfg --steps 10 --warmup 3 --no-cpu-baseline --launch eager 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('two_pass=$tp [$cfg]', 'step', d['ms_per_step'], 'A', d['paths']['epipolar_only_ms_per_step'], 'fgrad', k['epipolar_feature_grad'], 'attn_bwd', k['epipolar_attention_backward'])"
done; done
