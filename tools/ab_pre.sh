python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-probes --launch eager 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('step', d['ms_per_step'], 'B', d['paths']['raster_only_ms_per_step'], 'pre_fwd', k['preprocess_forward'], 'pre_bwd', k['preprocess_backward'], 'sort', k['depth_sort'])"
