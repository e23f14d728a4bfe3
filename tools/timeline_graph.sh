export TMPDIR=/tmp
rm -rf /tmp/tlg
rocprofv3 --kernel-trace --output-format csv -d /tmp/tlg -o p -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-probes > /tmp/tlg.log 2>&1
f=$(find /tmp/tlg -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY' > gpurun_out/r6tlg_timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last three occurrences of tiles_backward; print from the one before last to the last (one full step)
idx = [i for i, r in enumerate(rows) if 'tiles_backward' in r['Kernel_Name']]
lo, hi = idx[-2], idx[-1]
t0 = int(rows[lo]['Start_Timestamp'])
prev_end = {}
last_end = None
for r in rows[lo:hi + 1]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    gap = (s - last_end) / 1e3 if last_end is not None else 0.0
    print(f"{s/1e3:9.1f} {(e-s)/1e3:8.1f} us  gap {gap:7.1f}  q{r.get('Queue_Id','?'):>3s}  {r['Kernel_Name'][:60]}")
    last_end = max(last_end or 0, e)
PY
wc -l gpurun_out/r6tlg_timeline.txt
