#!/usr/bin/env python3
"""The last launches of a rocprofv3 --kernel-trace CSV as a timeline: start, end, duration, kernel, queue — which kernel
waited for which (the gaps between two mixes are read off here).
usage: tools/timeline.py <kernel_trace.csv> [launches=24]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("wbx::", "")[:40],
             r["Queue_Id"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["VGPR_Count"], r["LDS_Block_Size"]) for r in rows)
t0 = ks[0][0]
prev_mix_end = None
for s, e, name, q, wgx, vg, lds in ks[-n:]:
    gap = ""
    if name.startswith("mix_kernel"):
        if prev_mix_end is not None:
            gap = f"  <- {1e-3 * (s - prev_mix_end):.1f} us after the previous mix"
        prev_mix_end = e
    print(f"{(s - t0) / 1e6:11.3f} {(e - t0) / 1e6:11.3f} ms {(e - s) / 1e3:9.1f} us  q{q} {name:40s} wgs.x {wgx:6d} vgpr {vg:>3s} lds {lds:>6s}{gap}")
