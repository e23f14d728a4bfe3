cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 240 bash tools/timeline.sh r4g -- > /dev/null 2>&1
grep -n "sort_" gpurun_out/r4g_timeline.txt | tail -12
timeout 300 python -m pytest tests -m gpu -q -x -k "raster_gpu or config1_256 or fourth" 2>&1 | tail -3
