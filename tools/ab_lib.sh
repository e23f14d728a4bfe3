# A/B of two builds of the library in ONE box: pixelsplat_amd/libpixelsplat_hip_base.so (copy of an
# earlier build) against the current one.  usage: tools/ab_lib.sh <tag> [bench args]
tag=${1:-ab_lib}; shift
for which in base new base new; do
  if [ $which = base ]; then export PIXELSPLAT_HIP_LIB=$PWD/pixelsplat_amd/libpixelsplat_hip_base.so; else unset PIXELSPLAT_HIP_LIB; fi
  python bench.py $* --steps 20 --warmup 3 --no-cpu-baseline --launch eager 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('[$which]', 'step', d['ms_per_step'], 'A', d['paths']['epipolar_only_ms_per_step'], 'B', d['paths']['raster_only_ms_per_step'], {n: round(v,4) for n,v in k.items() if v > 0.03})"
done 2>&1 | tee gpurun_out/$tag.log
