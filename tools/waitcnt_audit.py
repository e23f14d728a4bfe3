"""Static audit of the compiled kernels for serialized memory round trips: counts, per kernel,
the s_waitcnt vmcnt(N) instructions that drain the queue (N small) right after few loads --
"load -> wait -> load -> wait" chains the source did not intend.  Usage:
    python tools/waitcnt_audit.py file.s [file.s ...]
Prints per kernel: loads, waits, and `chains` = waits with vmcnt <= 1 that have at most 2 loads
since the previous wait (each is one exposed round trip per execution of that code)."""
import re
import sys

for path in sys.argv[1:]:
    name, rows = None, []
    loads = waits = chains = since = 0
    depth_note = []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
            loads = waits = chains = since = 0
            continue
        if name is None:
            continue
        t = line.strip()
        if t.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
            loads += 1
            since += 1
        m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", t)
        if m:
            waits += 1
            if int(m.group(1)) <= 1 and 0 < since <= 2:
                chains += 1
            since = 0
        if t.startswith(".Lfunc_end"):
            rows.append((chains, loads, waits, name))
            name = None
    for chains, loads, waits, name in sorted(rows, reverse=True):
        if chains:
            print(f"{chains:4d} chains  {loads:4d} loads {waits:4d} waits  {name[:110]}")
