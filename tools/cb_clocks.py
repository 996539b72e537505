#!/usr/bin/env python3
"""tools/cb_clocks.py — where does a one-launch callback block spend its time?  (WBX_CB_DBG=1 diagnostic of libwbx:
every workgroup of callback_kernel notes the wall clock at start / sequencer done / mix done / ticket taken, the last one at
the end)   usage: python tools/cb_clocks.py [workload=c3] [tracks=4096]"""
import ctypes as C
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ["WBX_CB_DBG"] = "1"
WL = sys.argv[1] if len(sys.argv) > 1 else "c3"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
import numpy as np  # noqa: E402
import whitebox_amd as W  # noqa: E402
from whitebox_amd import synth  # noqa: E402

eng, seed, amp = b.build_device_session(W, synth, WL, N, 1, 64, 0, 1, 0)
out = W.AudioBuffer(512, 2)
eng.play()
L = W.lib()
L.wbx_debug_wg_clocks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
rows = []
for it in range(12):
    eng.process(None, out, 48000.0)
    n = C.c_size_t()
    L.wbx_debug_wg_clocks(eng.ctx.h, None, 0, C.byref(n))
    buf = np.zeros(4 * n.value, np.uint64)
    L.wbx_debug_wg_clocks(eng.ctx.h, buf.ctypes.data, buf.size, C.byref(n))
    if it < 4:
        continue
    ng, _, _ = eng.ctx.render_order(1)
    rec = buf[:6 * ng].reshape(ng, 6).astype(np.int64)
    t0 = rec[:, 0].min()
    us = lambda col: (rec[:, col] - t0) / 100.0
    last = int(np.argmax(rec[:, 4]))
    rows.append((us(0).max(), np.median(us(1) - us(0)), (us(1) - us(0)).max(), np.median(us(2) - us(1)), (us(2) - us(1)).max(),
                 us(2).max(), us(3).max(), (rec[last, 4] - t0) / 100.0, (rec[last, 4] - rec[last, 3]) / 100.0))
r = np.array(rows)
print(f"{WL} N={N} {eng.ctx.kernel_name()} groups {ng}: medians over {len(r)} blocks, microseconds from the first workgroup's start")
for name, col in (("last workgroup starts", 0), ("sequencer: median workgroup", 1), ("sequencer: slowest", 2), ("mix: median workgroup", 3),
                  ("mix: slowest", 4), ("last mix done at", 5), ("last ticket at", 6), ("flag out at (kernel span)", 7), ("sum + stores (last workgroup)", 8)):
    print(f"  {name:34s} {np.median(r[:, col]):7.2f}")
