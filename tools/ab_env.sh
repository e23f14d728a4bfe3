#!/bin/bash
# A/B of env-selected kernel variants in ONE box: each argument after the tag is an env assignment
# list ("PS_X=1 PS_Y=2", "" = defaults); every variant runs the configs[1] bench step twice
# (interleaved) and prints the step and kernel-group times.
# usage: tools/ab_env.sh <tag> "" "PS_COLOR_FWD_GPW=16" ...       [BENCH_ARGS="--size 512 --batch 2"]
tag=$1; shift
out=gpurun_out/$tag.log; : > $out
for rep in 1 2; do
  for envs in "$@"; do
    env $envs timeout 240 python bench.py ${BENCH_ARGS:-} --steps 20 --warmup 3 --no-cpu-baseline --no-probes 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']; n=d['kernel_launches_per_step']
print('[%s]' % '$envs', 'step', d['ms_per_step'], 'A', d['paths']['epipolar_only_ms_per_step'], 'B', d['paths']['raster_only_ms_per_step'], {g: round(v,4) for g,v in k.items() if v*n[g] > 0.03})" >> $out 2>&1
  done
done
cat $out
