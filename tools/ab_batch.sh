for b in 3 4 7 14 21; do
python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --launch eager 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('batch=$b tiles', $b*4*256, 'fwd', k['tiles_forward'], 'bwd', k['tiles_backward'], 'per-1k-tiles fwd %.4f bwd %.4f' % (k['tiles_forward']/($b*4*256)*1000, k['tiles_backward']/($b*4*256)*1000), 'D', d['config']['tile_list_entries_D'])"
done
