python -m pytest tests/test_epipolar_gpu.py tests/test_graph_gpu.py -m gpu -x -q 2>&1 | tail -2
for cfg in "" "--context-views 3 --batch 4" "--size 512 --batch 2"; do
python bench.py $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-probes --launch eager 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('[$cfg]', 'step', d['ms_per_step'], 'A', d['paths']['epipolar_only_ms_per_step'], 'fgrad', k['epipolar_feature_grad'])"
done
