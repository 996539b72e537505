#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace CSV (the table --stats prints), plus the timeline of
one steady-state step (gaps between consecutive kernels).
usage: tools/kt_stats.py <kernel_trace.csv> [--timeline]"""
import csv
import sys


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    agg = {}
    for r in rows:
        n = r["Kernel_Name"]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg.setdefault(n, [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1.0
    print(f"# {path}")
    print(f"{'kernel':<58} {'calls':>6} {'total_us':>11} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6}")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = n if len(n) <= 58 else n[:55] + "..."
        print(f"{short:<58} {a[0]:>6} {a[1]:>11.1f} {a[1] / a[0]:>9.2f} {a[2]:>9.2f} {a[3]:>9.2f} {100 * a[1] / total:>6.1f}")
    if "--timeline" in sys.argv:
        # last full step: from the last plan_kernel to the end
        idx = [i for i, r in enumerate(rows) if "plan_kernel" in r["Kernel_Name"]]
        if len(idx) >= 3:
            lo, hi = idx[-3], idx[-2]
            t0 = int(rows[lo]["Start_Timestamp"])
            print("# timeline of one step (us since its plan_kernel start; gap = idle before the kernel)")
            prev_end = None
            for r in rows[lo:hi + 1]:
                s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
                print(f"  {(s - t0) / 1e3:>9.2f}  +{(e - s) / 1e3:>8.2f}  gap {gap:>7.2f}  {r['Kernel_Name'][:60]}")
                prev_end = e


if __name__ == "__main__":
    main()
