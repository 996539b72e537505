#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean counter value per dispatch.
usage: tools/pmc_summary.py <dir with pass subdirs> [kernel substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "mix_kernel"
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print(f"# {os.path.relpath(f, root)}")
        for k, ctrs in acc.items():
            if want not in k:
                continue
            n = max(len(v) for v in ctrs.values())
            print(f"kernel: {k[:70]}  dispatches: {n}")
            for c, v in sorted(ctrs.items()):
                print(f"  {c:<32} mean/dispatch {sum(v) / len(v):>18.1f}")


if __name__ == "__main__":
    main()
