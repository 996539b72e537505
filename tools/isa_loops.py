#!/usr/bin/env python3
"""Innermost loops of one kernel in a hipcc -S dump: instruction-class histogram per loop body.
usage: tools/isa_loops.py <file.s> <kernel symbol substring> [min VALU instructions]"""
import re
import sys
from collections import Counter


def main():
    path, want = sys.argv[1], sys.argv[2]
    min_valu = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and want in l and l.rstrip().endswith(":") is False and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB[0-9_]+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
    for a, b in inner:
        ops = [l.split()[0] for l in body[a:b + 1] if l.startswith("\t") and not l.strip().startswith((";", "."))]
        valu = [o for o in ops if o.startswith("v_")]
        if len(valu) < min_valu:
            continue
        c = Counter(ops)
        cls = Counter()
        for o, n in c.items():
            k = ("f64" if "f64" in o and not o.startswith("v_cvt") else "cvt" if o.startswith("v_cvt") else "readlane" if "readlane" in o or "readfirstlane" in o
                 else "valu" if o.startswith("v_") else "salu" if o.startswith("s_") else "vmem" if o.startswith(("global_", "buffer_", "flat_")) else "lds" if o.startswith("ds_") else "other")
            cls[k] += n
        print(f"loop lines {start + a + 1}-{start + b + 1}: {len(ops)} instrs, VALU {len(valu)}", dict(cls))
        print("   ", ", ".join(f"{o}:{n}" for o, n in c.most_common(40)))


if __name__ == "__main__":
    main()
