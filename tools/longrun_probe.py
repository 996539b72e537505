#!/usr/bin/env python3
"""tools/longrun_probe.py — does the mix kernel's launch time HOLD?  Seconds of consecutive steps of the headline
workload, the HIP-event kernel time and the wall time per step printed for every chunk of steps: clocks settling
(the first ~20 ms), power / thermal drift afterwards, transport rewinds at the end of the resident session.
usage: tools/longrun_probe.py [workload=c3] [blocks per step=1024] [seconds=4]   -> profiles/rNN_longrun_probe.txt"""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
import whitebox_amd as W  # noqa: E402
from whitebox_amd import synth  # noqa: E402
from whitebox_amd.dist import PinnedBuffer  # noqa: E402

WL = sys.argv[1] if len(sys.argv) > 1 else "c3"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
SECONDS = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
n_tracks = 256 if WL == "c2" else 4096
per_block = n_tracks * 2 * b.FMT_BYTES[b.WORKLOADS[WL][3]] * 512 * b.WORKLOADS[WL][1] / 48000
SB = int(min(96e9 // per_block, 8 * K)) // K * K
eng, seed, amp = b.build_device_session(W, synth, WL, n_tracks, K, SB, 0, 1, 0)
host = PinnedBuffer(K * 2 * 512)
eng.ctx.set_master_target(host.ptr)
alg = b.algorithmic_bytes_per_block(n_tracks, b.WORKLOADS[WL][1], fmt=b.WORKLOADS[WL][3]) * K
eng.play()
done, chunk, steps_per_chunk = 0, 0, max(2, 8192 // K)
print(f"# {WL}, {n_tracks} tracks, {K} blocks per step, resident session {SB} blocks, {steps_per_chunk} steps per line")
t_start = time.perf_counter()
while time.perf_counter() - t_start < SECONDS:
    eng.ctx.kernel_time(reset=True)
    t0 = time.perf_counter()
    rew = 0
    for s in range(steps_per_chunk):
        if done + K > SB:
            eng.stop()
            eng.play()
            done = 0
            rew += 1
        eng.render(K)
        W.lib().wbx_pace(eng.ctx.h, 12)
        done += K
    eng.ctx.sync()
    dt = time.perf_counter() - t0
    ms, n = eng.ctx.kernel_time()
    print("t %6.3f s  steps %4d..%4d  step %.4f ms  mix %.4f ms  frac %.3f  frac_step %.3f  rewinds %d"
          % (time.perf_counter() - t_start, chunk * steps_per_chunk, (chunk + 1) * steps_per_chunk - 1, dt / steps_per_chunk * 1e3, ms,
             alg / (ms * 1e-3) / 8e12, alg / (dt / steps_per_chunk) / 8e12, rew), flush=True)
    chunk += 1
eng.close()
host.close()
