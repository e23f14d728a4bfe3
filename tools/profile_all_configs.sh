#!/bin/bash
# On the GPU box, in ONE call: the GPU test suite, tools/profile_bench.sh for the three benchmarked BASELINE
# configurations (bench line with the parity / CPU-baseline blocks, rocprofv3 kernel stats of the command and of the
# step alone, FETCH / WRITE and SQ counter summaries with the build stamp, roctx ranges), and for each of the other
# scene distributions (bench.py --scene) the bench line with its parity block + the step's kernel statistics
# -> gpurun_out/<tag>_*; copy what matters to profiles/.
# Round 5 additions: `comm` = the one-rank RCCL leg at configs[1] with the whole network's 0.48 GB of gradients
# as payload in both launch modes (exposed communication per step), `connected` = the bench line with the
# connected-step probe.
# usage: tools/profile_all_configs.sh [tag, default r6] [what: "tests c2 c4 c5 scenes comm connected", default all]
cd "$(dirname "$0")/.."
tag=${1:-r6}; what=${2:-"tests c2 c4 c5 scenes comm connected"}
mkdir -p gpurun_out
export TMPDIR=/tmp
has() { case " $what " in *" $1 "*) return 0;; esac; return 1; }
date
if has tests; then
  python -c "from pixelsplat_amd import _lib; print('build:', _lib.load().ps_build_info().decode())" > gpurun_out/${tag}_gpu_tests.log 2>/dev/null
  timeout -k 10 1200 python -m pytest tests -m gpu -q >> gpurun_out/${tag}_gpu_tests.log 2>&1; tail -3 gpurun_out/${tag}_gpu_tests.log; date
  timeout -k 10 200 python __graft_entry__.py --smoke > gpurun_out/${tag}_smoke.log 2>&1; tail -4 gpurun_out/${tag}_smoke.log
fi
has c2 && { timeout -k 10 1200 tools/profile_bench.sh ${tag}_c2; date; }
has c4 && { timeout -k 10 1200 tools/profile_bench.sh ${tag}_c4 --context-views 3 --batch 4; date; }
has c5 && { timeout -k 10 1200 tools/profile_bench.sh ${tag}_c5 --size 512 --batch 2; date; }
if has scenes; then
  for s in dense opaque large; do
    timeout -k 10 600 python bench.py --scene $s > gpurun_out/${tag}_scene_${s}_bench.json 2> gpurun_out/${tag}_scene_${s}_bench.err
    head -c 300 gpurun_out/${tag}_scene_${s}_bench.json; echo
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_sp_$s -o p -- python bench.py --scene $s --steps 10 --warmup 0 --no-cpu-baseline --no-probes --launch eager > gpurun_out/${tag}_sp_$s.log 2>&1
    f=$(ls gpurun_out/${tag}_sp_$s/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${tag}_scene_${s}_step_kernel_stats.csv
    rm -rf gpurun_out/${tag}_sp_$s
    date
  done
fi
if has comm; then
  for mode in eager auto; do
    PIXELSPLAT_FORCE_COMM=1 timeout -k 10 300 python bench.py --grad-payload-mb 480 --launch $mode --no-cpu-baseline --no-probes \
      > gpurun_out/${tag}_rccl_one_rank_c2_payload480_$mode.json 2> gpurun_out/${tag}_rccl_one_rank_$mode.err
    python -c "
import json; d=json.loads(open('gpurun_out/${tag}_rccl_one_rank_c2_payload480_$mode.json').read().strip().splitlines()[-1]); c=d['comm']
print('$mode', d['launch'], d['ms_per_step'], 'exposed', c['exposed_ms_per_step'], 'bytes', c['gradient_bytes_per_step'], c['extra_payload_bytes_per_step'], c.get('env'))"
  done; date
fi
if has connected; then
  timeout -k 10 600 python bench.py --connected --no-cpu-baseline > gpurun_out/${tag}_c2_bench_connected.json 2> gpurun_out/${tag}_c2_bench_connected.err
  python -c "
import json; d=json.loads(open('gpurun_out/${tag}_c2_bench_connected.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['paths']['connected'])"; date
fi
