#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_run.sh into profiles/pmc_traffic.json, the per-launch
HBM traffic of the dominant kernel that bench.py reports as roofline.traffic.

Corrections (MI355X_MICROARCH.md §HBM + own calibration, profiles/r01_fetch_calibration.txt):
  * FETCH_SIZE (KB) reports exactly half of the bytes of a wide coalesced streaming read on gfx950: x2.
    Calibrated on tools/ubench/read_bw.hip (known byte counts): contiguous float4 and the mix kernel's row
    pattern -> factor 2.000; the unaligned window pattern -> 2 x FETCH = 1.07 x unique bytes (line overlap).
  * WRITE_SIZE (KB) is used as reported: for the mix kernel it matches the known store volume
    (partial sums + peaks + level atomics) to within the atomics' share.
usage: tools/pmc_traffic.py <pmc dir> <workload> <K> <N> [kernel substring] [key suffix, e.g. _L5.3]"""
import csv
import glob
import json
import os
import sys


def mean_counter(root, counter, kernel):
    vals = []
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals) if vals else None


def main():
    root, workload, K, N = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    kernel = sys.argv[5] if len(sys.argv) > 5 else "mix_kernel"
    suffix = sys.argv[6] if len(sys.argv) > 6 else ""
    fetch = mean_counter(root, "FETCH_SIZE", kernel)
    write = mean_counter(root, "WRITE_SIZE", kernel)
    out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    data = json.load(open(out_path)) if os.path.exists(out_path) else {}
    data[f"{workload}_K{K}_N{N}{suffix}"] = {
        "kernel": kernel,
        "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB_raw": write,
        "fetch_bytes": fetch * 1024 * 2 if fetch is not None else None,
        "write_bytes": write * 1024 if write is not None else None,
        "hbm_bytes_per_launch": (fetch * 1024 * 2 + (write or 0) * 1024) if fetch is not None else None,
        "corrections": "FETCH_SIZE x2 (gfx950, calibrated); WRITE_SIZE as reported",
        "source": os.path.relpath(root, os.path.dirname(out_path) + "/.."),
    }
    json.dump(data, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(data[f"{workload}_K{K}_N{N}{suffix}"], indent=1))


if __name__ == "__main__":
    main()
