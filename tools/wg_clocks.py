#!/usr/bin/env python3
"""tools/wg_clocks.py — when do the workgroups of one mix launch start and end?  (WBX_DBG_CLOCK=1 diagnostic of libwbx)
usage: WBX_DBG_CLOCK=1 python tools/wg_clocks.py [workload=i16] [blocks=1024] [exact_min_blocks=1024|0]"""
import ctypes as C
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ["WBX_DBG_CLOCK"] = "1"
WL = sys.argv[1] if len(sys.argv) > 1 else "i16"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
if len(sys.argv) > 3:
    os.environ["WBX_EXACT_MIN_BLOCKS"] = sys.argv[3]
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
import numpy as np  # noqa: E402
import whitebox_amd as W  # noqa: E402
from whitebox_amd import synth  # noqa: E402

eng, seed, amp = b.build_device_session(W, synth, WL, 4096, K, 4 * K, 0, 1, 0)
eng.play()
for _ in range(3):
    eng.render(K)
eng.ctx.sync()
L = W.lib()
n = C.c_size_t()
L.wbx_debug_wg_clocks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
L.wbx_debug_wg_clocks(eng.ctx.h, None, 0, C.byref(n))
buf = np.zeros(4 * n.value, np.uint64)
L.wbx_debug_wg_clocks(eng.ctx.h, buf.ctypes.data, buf.size, C.byref(n))
rec = buf.reshape(-1, 4).astype(np.int64)
t0 = rec[:, 0].min()
start, end = (rec[:, 0] - t0) / 100.0, (rec[:, 1] - t0) / 100.0          # microseconds (100 MHz clock)
dur = end - start
hw, xcc = rec[:, 2], rec[:, 3] & 0xF
cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
print(f"{WL} K={K} {eng.ctx.kernel_name()}  workgroups {n.value}  kernel span {end.max():.0f} us")
print("  start  us: p0 %.0f p50 %.0f p90 %.0f p100 %.0f" % tuple(np.percentile(start, [0, 50, 90, 100])))
print("  end    us: p0 %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f p100 %.0f" % tuple(np.percentile(end, [0, 10, 50, 90, 99, 100])))
print("  length us: p0 %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f p100 %.0f  mean %.0f" % (*np.percentile(dur, [0, 10, 50, 90, 99, 100]), dur.mean()))
if n.value <= 4096:
    # where did the workgroups land?  (xcc, se, sh, cu) -> how many, how long
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    uniq, cnt = np.unique(key, return_counts=True)
    print(f"  CUs used {len(uniq)}; workgroups per CU: " + ", ".join(f"{c}: {np.sum(cnt == c)} CUs" for c in sorted(set(cnt))))
    for c in sorted(set(cnt)):
        sel = np.isin(key, uniq[cnt == c])
        print(f"    CUs holding {c} workgroups: mean length {dur[sel].mean():7.0f} us  max {dur[sel].max():7.0f} us")
    slow = np.argsort(-dur)[:12]
    print("  slowest workgroups (id, block x, xcc, se, sh, cu, length us, co-resident on its CU):")
    for i in slow:
        print(f"    {i:5d} {i % K:5d}  xcc {xcc[i]} se {se[i]} sh {sh[i]} cu {cu[i]:2d}  {dur[i]:7.0f}  {int(cnt[np.searchsorted(uniq, key[i])])}")
    for x in range(8):
        sel = xcc == x
        if sel.any():
            print(f"  XCC {x}: workgroups {sel.sum():5d}  CUs {len(np.unique(key[sel])):3d}  mean length {dur[sel].mean():8.0f} us  last end {end[sel].max():8.0f} us")
eng.close()
