#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per launch for kernels whose name contains
a pattern.     python3 tools/pmc_summary.py <dir> <kernel-substring> [...]"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    root, pats = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(root + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            for p in pats:
                if p in r["Kernel_Name"]:
                    acc[p][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    acc[p]["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for p in pats:
        print(f"== {p}")
        for name, vals in sorted(acc[p].items()):
            vals = vals[len(vals) // 3:] if len(vals) > 3 else vals        # drop warm-up launches
            print(f"   {name:34s} {sum(vals)/len(vals):16.1f}  (n={len(vals)})")


if __name__ == "__main__":
    main()
