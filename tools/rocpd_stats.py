#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (ROCm 7.2 default output) as the per-kernel table that
`--stats` would print: calls, total/avg/min/max duration (us), % of GPU kernel time.
usage: tools/rocpd_stats.py results.db [--skip-first N] > profiles/xxx_kernel_stats.txt"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    agg = {}
    for name, s, e in rows:
        d = (e - s) / 1e3
        a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1.0
    print(f"# {db}")
    print(f"{'kernel':<60} {'calls':>6} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) <= 60 else name[:57] + "..."
        print(f"{short:<60} {a[0]:>6} {a[1]:>12.1f} {a[1] / a[0]:>10.2f} {a[2]:>10.2f} {a[3]:>10.2f} {100 * a[1] / total:>6.1f}")


if __name__ == "__main__":
    main()
