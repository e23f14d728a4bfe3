# usage: tools/kstats.sh <tag> [bench args]: rocprofv3 kernel stats of a short eager bench run (step only), top kernels
tag=$1; shift
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -o p -- python bench.py $* --steps 10 --warmup 0 --no-cpu-baseline --no-probes --launch eager > gpurun_out/${tag}_prof.log 2>&1
f=$(ls gpurun_out/${tag}_prof/*kernel_stats.csv | head -1); cp $f gpurun_out/${tag}_kernel_stats.csv; rm -rf gpurun_out/${tag}_prof
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${tag}_kernel_stats.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:60]:
    print(f'{float(r["TotalDurationNs"])/1e6:8.2f} ms  n={int(r["Calls"]):4d}  avg={float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:110]}')
PY
