"""Summarise rocprofv3 --pmc SQ_* passes of bench.py (mean per launch, per kernel).

    for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \\
               "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \\
               "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
      rocprofv3 --kernel-trace --pmc $set --output-format csv -d out/$n -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
    done
    python tools/pmc_sq_summary.py out/*/p_counter_collection.csv > profiles/rN_pmc_sq.json

SQ_ACTIVE_INST_VALU counts quad-cycles (one wave64 VALU instruction = one count = 4 cycles of
a SIMD), so  valu_busy_ms = ACTIVE_INST_VALU * 4 / (1024 SIMDs * 2.4e6 cycles/ms).
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sys.argv[1:]:
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "ps::" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, d in sorted(agg.items()):
        row = {c: round(sum(v) / len(v)) for c, v in d.items()}
        if "SQ_ACTIVE_INST_VALU" in row:
            row["valu_busy_ms_at_2.4GHz"] = round(row["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * 2.4e6), 4)
        out[k] = row
    from build_stamp import stamp
    json.dump(dict(unit="mean counter value per launch", **stamp(), kernels=out), sys.stdout, indent=1)


if __name__ == "__main__":
    main()
