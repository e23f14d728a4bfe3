#!/usr/bin/env python3
"""Static census of a kernel's gfx950 ISA: VALU / SALU / LDS / memory instructions per basic block with the
loop nest the assembler comments give -- how `profiles/r4_attention_instruction_census.txt` was counted.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ipixelsplat_amd/csrc --cuda-device-only -S \
          pixelsplat_amd/csrc/epipolar_attention.hip -o /tmp/ea.s
    python tools/isa_census.py /tmp/ea.s epipolar_attn_forward_kernelILi128ELb0ELi1E [--min 8] [--list LBB17_36]

The second argument is any substring of the mangled kernel name (first match).  A row runs from one label to the
next: fall-through blocks the compiler did not label count with their predecessor.  Dynamic counts are the reader's
job: blocks of a loop times its trip count (for the attention kernels: one trip of the depth-1 loop = one chunk of
8 tokens; the depth-2 loop of the context phase runs once per token)."""
from __future__ import annotations

import argparse
import collections
import re


def blocks_of(path: str, name: str):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and name in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    out, cur = [], ["entry", [], ""]
    out.append(cur)
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?", l)
        if m:
            cur = [m.group(1), [], m.group(2) or ""]
            out.append(cur)
            continue
        t = l.strip()
        if not t or t.startswith((";", ".")):
            if "Loop" in t:
                cur[2] += " " + t
            continue
        cur[1].append(t)
    return lines[start].split(":")[0], out


def kind(op: str) -> str:
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("kernel")
    ap.add_argument("--min", type=int, default=1, help="hide blocks with fewer VALU instructions")
    ap.add_argument("--list", default=None, help="print the instructions of this block (label without the dot)")
    args = ap.parse_args()
    full, blocks = blocks_of(args.asm, args.kernel)
    print(f"# {full}")
    print("%-12s %5s %5s %5s %4s %5s  %s" % ("block", "instr", "valu", "salu", "lds", "vmem", "loop nest"))
    tot = collections.Counter()
    for label, instrs, tag in blocks:
        c = collections.Counter(kind(x.split()[0]) for x in instrs)
        tot.update(c)
        if c["valu"] >= args.min:
            nest = re.sub(r"\s+", " ", tag.replace(";", " ")).strip()
            print("%-12s %5d %5d %5d %4d %5d  %s" % (label, len(instrs), c["valu"], c["salu"], c["lds"], c["vmem"],
                                                     nest[:110]))
        if args.list and label.lstrip(".") == args.list:
            for x in instrs:
                print("        " + x[:110])
    print("%-12s %5d %5d %5d %4d %5d  (static totals)" % ("all", sum(tot.values()), tot["valu"], tot["salu"],
                                                         tot["lds"], tot["vmem"]))


if __name__ == "__main__":
    main()
