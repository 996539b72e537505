#!/usr/bin/env python3
"""The numbers table of DESIGN.md §5 from the committed bench lines (profiles/r02_bench_*.json): prints the markdown rows.
usage: tools/design_table.py"""
import json
import os

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")


def L(w):
    return json.load(open(os.path.join(ROOT, f"r02_bench_{w}.json")))


def cells(w, kernel=True):
    d = L(w)
    r = d["roofline"]
    k = r["kernel"].replace("wbx::", "").replace(", ", ",")
    tr = f"{r['traffic'] / r['algorithmic_bytes_per_launch']:.2f}" if r.get("traffic") else "—"
    return dict(v=f"{d['value']:.3g}".replace("e+0", "e"), step=f"{d['ms_per_step']:.3f}", k=f"`{k}`" if kernel else "",
                ms=f"{r['kernel_ms_avg']:.3f} ms", gbs=f"{r['achieved']:.0f}", frac=f"{r['frac']:.2f}", tr=tr, val=d["value"])


def row(label, w, note="", frac_note="", kernel=True):
    c = cells(w, kernel)
    return f"| {label} | {c['v']}{note} | {c['step']} | {c['k']} {c['ms']} | {c['gbs']} | {c['frac']}{frac_note} | {c['tr']} |"


def pair(label, a, b, base, note):
    ca, cb, cbase = cells(a), cells(b), cells(base)
    da, db = 100 * (ca["val"] / cbase["val"] - 1), 100 * (cb["val"] / cbase["val"] - 1)
    return (f"| {label} | {ca['v']} (**{da:+.0f} %**) / {cb['v']} (**{db:+.0f} %**){note} | {ca['step']} / {cb['step']} | "
            f"{ca['ms']} / {cb['ms']} | {ca['gbs']} / {cb['gbs']} | {ca['frac']} / {cb['frac']} | — |")


print("| workload | frames/s (`value`) | ms/step | dominant kernel, mean launch (HIP events) | achieved GB/s | frac of 8 TB/s | PMC traffic ÷ algorithmic |")
print("|---|---|---|---|---|---|---|")
print(row("**c3** (configs[2], headline)", "default", " (1.62-1.76e8 over the runs and boxes of the round)", " (0.65-0.70 over the runs and boxes)"))
print(row("c4 (configs[3], 64 buses)", "c4", frac_note=" (0.70-0.76)"))
print(row("c2 (configs[1], 256 tracks)", "c2", frac_note=" (2 workgroup columns × 256 blocks: launch- and tail-bound; smaller groups and the other instances are slower, `tools/ab_c2_groups.sh`)"))
print(pair("c3 cut into clips of 5.3 / 20 blocks (a clip boundary in 19 % / 5 % of all track-blocks)", "c3_L5.3", "c3_L20", "default", "; round 1: −50 % / −14 %"))
print(row("i16 (c4 with 16-bit clips)", "i16"))
print(pair("i16 cut into clips of 5.3 / 20 blocks", "i16_L5.3", "i16_L20", "i16", "; through the pre-render pass, as until this round: −70 % / −45 %"))
print(row("mixfmt (fp32 / 16-bit / 24-bit per track, session rate; `MODE_MU`)", "mixfmt"))
print(row("d96 (96 kHz clips at speed 2, per-frame taps; `MODE_G`)", "d96"))
print(row("i24r (24-bit 44.1 kHz clips; `MODE_WN`, fp64 normalisation)", "i24r"))
print(row("mixr (16-bit 44.1 kHz alternating with 24-bit 48 kHz; `MODE_MWN`)", "mixr"))
print(row("i16r (16-bit 44.1 kHz clips; `MODE_WINU`)", "i16r", frac_note=" — **issue-bound, not HBM-bound** (below)"))
print(pair("i16r cut into clips of 5.3 / 20 blocks", "i16r_L5.3", "i16r_L20", "i16r", "; through the pre-render pass, as until this round: −65 %"))
a, b = cells("dist1_reduce"), cells("dist1_ordered")
print(f"| multi-GPU code path on one rank, real RCCL (`--force-dist-path`: reduce / ordered) | {a['v']} / {b['v']} | {a['step']} / {b['step']} | {a['ms']} / {b['ms']} | {a['gbs']} / {b['gbs']} | {a['frac']} / {b['frac']} | — |")
