#!/usr/bin/env python3
"""The committed per-kernel summary (profiles/rNN_c3_K256_kernel_summary.txt) from rocprofv3 --kernel-trace CSVs
of the bench command: launches, mean over all launches, mean over the last `timed` launches (= the timed
steps), min, max — next to the HIP-event figure of the same run's bench line.
usage: tools/kernel_summary.py <trace.csv> <bench.json> <title> [<trace.csv> <bench.json> <title> ...]"""
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"wbx::(\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "").replace(" ", "")) if m else None


ORDER = ["times_copy_kernel", "plan_kernel", "plan_kernel_beside", "plan_seg_kernel", "plan_seg_kernel_beside", "gen_kernel", "mix_kernel", "sum_kernel"]


def one(trace, bench, title):
    d = json.loads([l for l in open(bench) if l.startswith("{")][-1])
    timed = d["steps"]
    rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
    per = {}
    for r in rows:
        k = short(r["Kernel_Name"])
        if k and k.split("<")[0] in ORDER:
            per.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"# rocprofv3 --kernel-trace of `{title}` ({d['config']['workload'].split(' ')[0]}, K={d['config']['blocks_per_step']}): "
          f"{d['warmup']} warm-up + {d['ramp_steps']} ramp + {timed} timed steps")
    print(f"# the same run's bench line: ms_per_step {d['ms_per_step']:.3f}, roofline.kernel_ms_avg (HIP events, timed steps) "
          f"{1e3 * d['roofline']['kernel_ms_avg']:.1f} us"
          + (f", over all {d['roofline']['kernel_launches_all']} launches {1e3 * d['roofline']['kernel_ms_avg_all_launches']:.1f} us"
             if "kernel_ms_avg_all_launches" in d["roofline"] else ""))
    print("# kernel, launches, mean us (all), mean us (last %d = the timed steps), min us, max us" % timed)
    for k in sorted(per, key=lambda k: ORDER.index(k.split("<")[0])):
        v = per[k]
        last = v[-timed:]
        print(f"{k:<34}{len(v):>6}{sum(v) / len(v):>11.1f}{sum(last) / len(last):>11.1f}{min(v):>11.1f}{max(v):>11.1f}")
    print()


a = sys.argv[1:]
for i in range(0, len(a), 3):
    one(a[i], a[i + 1], a[i + 2])
