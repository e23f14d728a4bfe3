for fb in "2 2" "2 4"; do set -- $fb
PS_TILES_FWD_VARIANT=$1 PS_TILES_BWD_VARIANT=$2 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --launch eager 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fwd=$1 bwd=$2', d['ms_per_step'], d['kernels_ms']['tiles_forward'], d['kernels_ms']['tiles_backward'])"
done
PS_TILES_BWD_VARIANT=4 python -m pytest tests/test_raster_gpu.py -m gpu -x -q 2>&1 | tail -2
