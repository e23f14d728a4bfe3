#!/bin/bash
# The ONE way GPU evidence is taken: stamps the tree's commit into .git_sha (the GPU box has
# no .git; tools/build_stamp.py reads it there), rebuilds the library if a source changed, and hands the command
# to gpurun.  Whatever the command writes under gpurun_out/ comes back; copy what is to be judged to profiles/.
#   tools/gpu_call.sh [--timeout SECONDS] -- '<command run from the repo root on the GPU box>'
cd "$(dirname "$0")/.."
sha=$(git rev-parse HEAD)
if [ -n "$(git status --porcelain -- pixelsplat_amd/csrc include bench.py pixelsplat_amd/*.py)" ]; then sha="$sha-dirty"; fi
echo "$sha" > .git_sha
python -m pixelsplat_amd.build > /dev/null || exit 1
exec /usr/local/graft/bin/gpurun "$@"
