cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -k "raster or decoder or scene or config or graph or bench" > gpurun_out/r4d_gpu_tests.log 2>&1; tail -4 gpurun_out/r4d_gpu_tests.log
timeout 600 tools/ab_variants.sh r4d_fwd_walk_ab "--steps 20 --warmup 3 --launch eager" r3:r3tiles walk0:walk0 walk1
