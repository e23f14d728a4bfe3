#!/bin/bash
# A/B of library variants (tools/build_variant.sh <tag> flags -> pixelsplat_amd/libps_<tag>.so) and of
# environment switches in ONE box, interleaved twice.  Each item is "label[:lib-tag][:ENV=VAL,...]";
# lib-tag "-" = the default library.   usage: tools/ab_variants.sh <out-tag> "<bench args>" item...
out=$1; shift; args=$1; shift
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
for rep in 1 2; do
for item in "$@"; do
  IFS=: read -r label lib envs <<< "$item"
  ( if [ -n "$lib" ] && [ "$lib" != "-" ]; then export PIXELSPLAT_HIP_LIB=$PWD/pixelsplat_amd/libps_$lib.so; fi
    if [ -n "$envs" ]; then for kv in ${envs//,/ }; do export "$kv"; done; fi
    python bench.py $args --no-cpu-baseline --no-probes 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']; p=d['paths']
print('[$label]', 'step', d['ms_per_step'], d['launch'], 'A', p.get('epipolar_only_ms_per_step'), 'B', p.get('raster_only_ms_per_step'), 'eager', p.get('eager_ms_per_step'), {n: round(v,4) for n,v in k.items() if v > 0.05})" )
done; done 2>&1 | tee gpurun_out/$out.log
