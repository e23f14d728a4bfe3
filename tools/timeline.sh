# kernel timeline (start, duration, stream/queue, name) of the last steps of a short bench run
# usage: tools/timeline.sh <tag> [env assignments as VAR=val ...] -- [bench args]
tag=$1; shift
export TMPDIR=/tmp
envs=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done; shift
rm -rf /tmp/tl_$tag
env "${envs[@]}" rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -o p -- python bench.py "$@" --steps 3 --warmup 2 --no-cpu-baseline --no-probes --launch eager > /tmp/tl_$tag.log 2>&1
f=$(find /tmp/tl_$tag -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY' > gpurun_out/${tag}_timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the whole last step: from the last camera_setup (the first kernel of (B)) to the end of the trace, (A) included
lo = max(i for i, r in enumerate(rows) if 'camera_setup' in r['Kernel_Name'])
t0 = int(rows[lo]['Start_Timestamp'])
for r in rows[lo:]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print(f"{s/1e3:9.1f} {(e-s)/1e3:8.1f} us  q{r.get('Queue_Id','?'):>3s}  {r['Kernel_Name'][:70]}")
PY
tail -n 40 gpurun_out/${tag}_timeline.txt
