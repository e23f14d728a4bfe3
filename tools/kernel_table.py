#!/usr/bin/env python3
"""One table per benchmarked configuration out of the three committed summaries of a
tools/profile_bench.sh run: rocprofv3 kernel stats, FETCH / WRITE counters, SQ counters.

    python tools/kernel_table.py r3_c2 [r3_c4 r3_c5 ...]      -> profiles/<tag>_kernel_table.txt

Algorithmic bytes: SURVEY.md 8(d)'s per-unit figures (the formulas bench.py prices its `roofline` block
with) for the kernels the formula names -- with the terms this design moves once per SCENE (the 340
B/Gaussian inputs, read by the forward and again by the backward, and the 340 B/Gaussian of gradients)
counted once per scene, not once per view as the contract's whole-path formula does (round 3's table
charged them per view and showed "fractions of the roofline" above 1 for three kernels).
"""
from __future__ import annotations

import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
HBM_PEAK = 8.0e12


def _short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)          # drop the argument list
    return name


def _alg_bytes(bench: dict) -> dict:
    """kernel-name fragment -> algorithmic bytes per launch, from the bench line's own workload."""
    cfg = bench["config"]
    G, V, D = cfg.get("gaussians_per_scene"), cfg.get("views_per_step_per_gpu"), cfg.get("tile_list_entries_D")
    m = re.search(r"(\d+)x(\d+), batch_size", cfg.get("workload", ""))
    if None in (G, V, D) or not m:
        return {}
    npix = int(m.group(1)) * int(m.group(2))
    mb = re.search(r"batch_size=(\d+)", cfg.get("workload", ""))
    S = int(mb.group(1)) if mb else V // 4       # scenes per step
    return {
        # inputs once per scene + the per-(view, Gaussian) state of SURVEY 8(d)
        "preprocess_fused_kernel": 340.0 * G * S + 52.0 * G * V,
        "tiles_forward_kernel": 36.0 * D + 20.0 * npix * V,
        "tiles_backward_kernel": 76.0 * D + 20.0 * npix * V,
        # inputs re-read and gradients written once per scene + the accumulated 2-D gradients per view
        "geometry_backward_kernel+color_backward_kernel": 680.0 * G * S + 48.0 * G * V,
    }


def table(tag: str) -> str:
    # the STEP-ONLY statistics when they exist (every launch of a kernel is then the benchmarked workload;
    # the statistics of the whole command also average over the parity block's and the probes' launches)
    stats_name = f"{tag}_step_kernel_stats.csv"
    if not os.path.exists(os.path.join(PROF, stats_name)):
        stats_name = f"{tag}_kernel_stats.csv"
    stats = list(csv.DictReader(open(os.path.join(PROF, stats_name))))

    def _load(name):     # counter summaries are optional (the scene runs take none)
        path = os.path.join(PROF, name)
        return json.load(open(path)) if os.path.exists(path) else {"kernels": {}}
    traffic, sq = _load(f"{tag}_pmc_traffic.json"), _load(f"{tag}_pmc_sq.json")
    bench_path = os.path.join(PROF, f"{tag}_bench.json")
    bench = json.loads(open(bench_path).read().strip().splitlines()[-1]) if os.path.exists(bench_path) else {}
    alg = _alg_bytes(bench) if bench else {}
    tkeys = {_short(k): v for k, v in traffic["kernels"].items()}
    skeys = {_short(k): v for k, v in sq["kernels"].items()}
    out = [
        f"# every ps:: kernel of the {tag} bench run: rocprofv3 average duration (profiles/{stats_name}),",
        f"# memory-side bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE,",
        f"# profiles/{tag}_pmc_traffic.json), the rate that makes, the algorithmic bytes of SURVEY.md 8(d) where the formula",
        f"# names the kernel, VALU instructions per launch and VALU-busy time (profiles/{tag}_pmc_sq.json).",
        f"# build: {traffic.get('build')}",
        f"# git:   {traffic.get('git')}        (tools/kernel_table.py {tag})",
        "%-58s %6s %8s %10s %6s %7s %9s %8s %8s" % ("kernel", "calls", "avg us", "traffic GB", "TB/s", "alg GB",
                                                     "alg/8TB/s", "VALU M", "busy us"),
    ]
    rows = [r for r in stats if "ps::" in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows:
        k = _short(r["Name"])
        avg_us = float(r["AverageNs"]) / 1e3
        t = tkeys.get(k)
        s = skeys.get(k)
        tb = t["bytes"] if t else float("nan")
        a = float("nan")
        for frag, val in alg.items():
            parts = frag.split("+")
            if any(p in k for p in parts):
                # a formula that covers two kernels is shown on each with the kernel's share of time
                if len(parts) > 1:
                    tot = sum(float(x["AverageNs"]) for x in rows if any(p in x["Name"] for p in parts))
                    val = val * float(r["AverageNs"]) / tot
                a = val
        out.append("%-58s %6d %8.1f %10.3f %6.2f %7.3f %9.3f %8.1f %8.1f" % (
            k[:58], int(r["Calls"]), avg_us, tb / 1e9, tb / (avg_us * 1e-6) / 1e12, a / 1e9,
            a / (avg_us * 1e-6) / HBM_PEAK,
            (s["SQ_INSTS_VALU"] / 1e6) if s else float("nan"),
            (s.get("valu_busy_ms_at_2.4GHz", float("nan")) * 1e3) if s else float("nan")))
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    for tag in sys.argv[1:] or ["r4_c2", "r4_c4", "r4_c5"]:
        text = table(tag)
        with open(os.path.join(PROF, f"{tag}_kernel_table.txt"), "w") as f:
            f.write(text)
        print(text)
