#!/usr/bin/env python3
"""tools/callback_calls.py <tracks> <calls> — `calls` consecutive wbx_engine_process calls (the audio callback) on a session of
`tracks` tracks of the bench's c3 workload; run under `rocprofv3 --kernel-trace --stats` it shows what one callback dispatches."""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
N, CALLS = int(sys.argv[1]), int(sys.argv[2])
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
import whitebox_amd as W  # noqa: E402
from whitebox_amd import synth  # noqa: E402

eng, _, _ = b.build_device_session(W, synth, "c3", N, 1, CALLS + 32, 0, 1, 0)
out = W.AudioBuffer(512, 2)
eng.play()
for _ in range(8):
    eng.process(None, out, 48000.0)
process, handle, ptrs = W.lib().wbx_engine_process, eng.h, out._ptrs()
t0 = time.perf_counter()
for _ in range(CALLS):
    if process(handle, ptrs) != 0:
        raise SystemExit("wbx_engine_process failed")
dt = (time.perf_counter() - t0) / CALLS
print(f"tracks {N}: {CALLS} calls of wbx_engine_process, {1e6 * dt:.2f} us per call, kernel {eng.ctx.kernel_name()}")
eng.close()
