# preprocess / colour kernels: kernel times of the step + raster parity tests
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --launch eager 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('step', d['ms_per_step'], 'B', d['paths']['raster_only_ms_per_step'], {n: k[n] for n in ('preprocess_forward','preprocess_backward','depth_sort','tile_bins')})"
bash tools/kstats.sh ${1:-pre2} > gpurun_out/${1:-pre2}_kstats.txt 2>&1; grep -E "color_|geometry_" gpurun_out/${1:-pre2}_kstats.txt | cut -c1-120
[ -n "$NOTEST" ] || python -m pytest tests/test_raster_gpu.py tests/test_raster_configs_gpu.py -m gpu -x -q 2>&1 | tail -2
