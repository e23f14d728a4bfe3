#!/bin/bash
# Profiles `python bench.py <args>` on the GPU box and leaves the summaries under gpurun_out/<tag>_*:
#   <tag>_bench.json           the bench line (full run, with the CPU baseline unless the args say no) -- taken
#                              LAST, after the counter summaries were copied into profiles/ ON THE BOX, so that
#                              its roofline.traffic / valu_issue_frac are paired with this very build
#   <tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats of a 10-step run of the whole bench command
#   <tag>_step_kernel_stats.csv  the same of the STEP ONLY (eager, no probes): the averages that agree with
#                              roofline.avg_kernel_ms (every launch of a kernel is the benchmarked workload)
#   <tag>_pmc_traffic.json     FETCH_SIZE / WRITE_SIZE per kernel (separate passes, gfx950 corrections)
#   <tag>_pmc_sq.json          SQ instruction / busy / wait counters per kernel (separate passes)
# usage: tools/profile_bench.sh r2x [--size 512 --batch 2 ...]      (copy what matters to profiles/)
set -u
tag=$1; shift
args="$*"
out=gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
short="$args --steps 10 --warmup 2 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o p -- python bench.py $short > $out/${tag}_prof.log 2>&1
f=$(ls $out/${tag}_prof/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $out/${tag}_kernel_stats.csv; else
  db=$(ls $out/${tag}_prof/*.db 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_kernel_stats.py "$db" > $out/${tag}_kernel_stats.csv; fi
# (--no-probes: the dense-scene probe launches the same kernels on ANOTHER workload; its launches would be averaged
# into the per-launch counters -- rounds 3 - 6 had them in: tiles_backward 824 M instead of 757 M VALU instructions)
pmc="$args --steps 2 --warmup 1 --no-cpu-baseline --no-probes"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/${tag}_pmc$i -o p -- python bench.py $pmc > $out/${tag}_pmc$i.log 2>&1
done
python tools/pmc_summary.py $out/${tag}_pmc1/*counter_collection.csv $out/${tag}_pmc2/*counter_collection.csv > $out/${tag}_pmc_traffic.json 2>> $out/${tag}_prof.log
python tools/pmc_sq_summary.py $out/${tag}_pmc3/*counter_collection.csv $out/${tag}_pmc4/*counter_collection.csv > $out/${tag}_pmc_sq.json 2>> $out/${tag}_prof.log
rm -rf $out/${tag}_pmc[1-4] $out/${tag}_prof   # raw traces are large; the summaries stay
cp $out/${tag}_pmc_traffic.json $out/${tag}_pmc_sq.json profiles/ 2>/dev/null
timeout 600 python bench.py $args > $out/${tag}_bench.json 2> $out/${tag}_bench.err
head -c 400 $out/${tag}_bench.json; echo
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_sprof -o p -- python bench.py $args --steps 10 --warmup 0 --no-cpu-baseline --no-probes --launch eager > $out/${tag}_sprof.log 2>&1
f=$(ls $out/${tag}_sprof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_step_kernel_stats.csv
rm -rf $out/${tag}_sprof
# roctx ranges of the same step as rocprofv3 --marker-trace sees them (kernel trace beside it, no counters)
PS_ROCTX=1 timeout 200 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $out/${tag}_roctx -o p -- python bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-probes --launch eager > $out/${tag}_roctx.log 2>&1
f=$(ls $out/${tag}_roctx/*marker_api_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_roctx_ranges.csv
rm -rf $out/${tag}_roctx
