"""Kernel statistics (the columns of rocprofv3's kernel_stats.csv) from a rocpd results .db.

    python tools/rocpd_kernel_stats.py gpurun_out/prof/x_results.db > profiles/rN_kernel_stats.csv
"""
import sqlite3
import statistics
import sys


def main(path):
    con = sqlite3.connect(path)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = {}
    for n, start, end in con.execute(f"select {name}, start, end from kernels"):
        rows.setdefault(n, []).append(end - start)
    total = sum(sum(v) for v in rows.values())
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"')
    for n, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        sd = statistics.stdev(v) if len(v) > 1 else 0.0
        print(f'"{n}",{len(v)},{sum(v)},{sum(v) / len(v):.6f},{100 * sum(v) / total:.2f},{min(v)},{max(v)},{sd:.6f}')


if __name__ == "__main__":
    main(sys.argv[1])
