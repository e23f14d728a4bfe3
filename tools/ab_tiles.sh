for f in 0 1; do for b in 0 1; do
PS_TILES_FWD_VARIANT=$f PS_TILES_BWD_VARIANT=$b python bench.py --steps 20 --warmup 3 --no-cpu-baseline --launch eager 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fwd=$f bwd=$b', d['ms_per_step'], d['kernels_ms']['tiles_forward'], d['kernels_ms']['tiles_backward'])"
done; done
python -m pytest tests/test_raster_gpu.py tests/test_raster_configs_gpu.py -m gpu -x -q 2>&1 | tail -3
