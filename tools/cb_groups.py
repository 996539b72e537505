#!/usr/bin/env python3
"""tools/cb_groups.py — callback latency (wbx_engine_process, one 512-frame block per call) against the track-group size of
a max_blocks = 1 engine, for a range of session sizes.   usage: python tools/cb_groups.py [workload=c3] [calls=800]"""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
WL = sys.argv[1] if len(sys.argv) > 1 else "c3"
CALLS = int(sys.argv[2]) if len(sys.argv) > 2 else 800
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
import whitebox_amd as W  # noqa: E402
from whitebox_amd import synth  # noqa: E402


def latency(tracks, group):
    eng, _, _ = b.build_device_session(W, synth, WL, tracks, 1, CALLS + 16, 0, 1, group)
    out = W.AudioBuffer(512, 2)
    eng.play()
    for _ in range(8):
        eng.process(None, out, 48000.0)
    process, handle, ptrs = W.lib().wbx_engine_process, eng.h, out._ptrs()
    runs, per = [], CALLS // 8
    for _ in range(8):
        t1 = time.perf_counter()
        for _ in range(per):
            if process(handle, ptrs) != 0:
                raise RuntimeError("wbx_engine_process failed")
        runs.append((time.perf_counter() - t1) / per)
    ng = eng.ctx.render_order(1)[0]
    name = eng.ctx.kernel_name().split("<")[0]
    eng.close()
    return 1e6 * sorted(runs)[4], ng, name


print(f"{WL}: median of 8 runs of {CALLS // 8} calls, microseconds per block (workgroups)")
for n in (8, 16, 24, 32, 48, 64, 65, 128, 256, 512, 1024, 2048, 4096):
    cells = []
    for g in (0, 64, 32, 16, 8, 4, 2, 1):
        if g and (-(-n // g) > 256 or (g >= n and g != 64)):
            cells.append(f"g{g}: -")
            continue
        us, ng, name = latency(n, g)
        cells.append(f"g{g or 'auto'}: {us:5.1f} ({ng}{'' if 'callback' in name else ' 3L'})")
    print(f"N={n:5d}  " + "  ".join(cells), flush=True)
