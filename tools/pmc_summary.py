"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of bench.py into one small JSON.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/f -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/w -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/pmc_summary.py out/f/p_counter_collection.csv out/w/p_counter_collection.csv > profiles/rN_pmc_traffic.json

Counters are KB per launch (mean).  Calibration on this machine (MI355X_MICROARCH.md, HBM
section, and our own known-size kernels): FETCH_SIZE reports half the bytes of 16-byte-per-lane
reads (colour forward: 431 811 KB reported for 826 MB of SH + 44 MB of radii), WRITE_SIZE is
1:1 (colour backward: 838 656 KB for 826 MB of dL/dSH).  `bytes` applies that: 2 x fetch + write.
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

def load(path):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in per.items()}


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        if "ps::" not in k:
            continue
        f, n = fetch.get(k, (0.0, 0))
        w, _ = write.get(k, (0.0, 0))
        kernels[k.replace("void ", "")] = dict(
            fetch_kb=round(f, 1), write_kb=round(w, 1), launches_profiled=n,
            bytes=round((2.0 * f + w) * 1024.0))
    from build_stamp import stamp
    json.dump(dict(unit="mean KB per launch; bytes = (2 x fetch_kb + write_kb) x 1024",
                   **stamp(), kernels=kernels), sys.stdout, indent=1)


if __name__ == "__main__":
    main()
