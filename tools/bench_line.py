#!/usr/bin/env python3
"""One bench.py JSON line (stdin) as one short text line: value, step and kernel time, both roofline fractions.
usage: python bench.py ... | tools/bench_line.py [label]"""
import json
import sys

for ln in sys.stdin:
    if not ln.startswith("{"):
        continue
    d = json.loads(ln)
    r = d["roofline"]
    v = d.get("verify")
    print("%-44s %.4g frames/s  step %.4f ms  mix %.4f ms  frac %.3f  frac_step %.3f  gap %.1f us  enqueue max %.3f ms%s  %s" % (
        " ".join(sys.argv[1:]), d["value"], d["ms_per_step"], r["kernel_ms_avg"], r["frac"], r["frac_step"], 1e3 * r.get("mix_gap_ms_avg", 0.0), d["host_enqueue_ms_max"],
        "" if v is None else ("  verify ok" if v.get("ok") else "  VERIFY FAILED"), r["kernel"].replace("wbx::", "")))
