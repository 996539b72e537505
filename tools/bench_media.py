#!/usr/bin/env python3
"""Throughput of the two rows next to the mix path (clip ingest, waveform mip-maps) on one MI355X.
Wall-clock per call here; the kernel durations come from `rocprofv3 --kernel-trace --stats` of this command
(profiles/r01_media_kernel_stats.csv).  Algorithmic bytes: ingest = 2 x frames x channels x elem (read + write);
mip-maps = frames x elem read + sum of the levels' outputs written, per channel."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import whitebox_amd as W  # noqa: E402

FRAMES = int(os.environ.get("FRAMES", 64 << 20))
ITERS = 5
out = []
ctx = W.MixContext(8)
for fmt, dt in (("i16", torch.int16), ("f32", torch.float32)):
    eb = 2 if fmt == "i16" else 4
    src = torch.randint(-30000, 30000, (FRAMES, 2), dtype=torch.int16, device="cuda") if fmt == "i16" \
        else torch.rand((FRAMES, 2), dtype=torch.float32, device="cuda") - 0.5
    torch.cuda.synchronize()
    ctx.clip_ingest_device(0, fmt, 2, 48000, FRAMES, src.data_ptr())
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(ITERS):
        ctx.clip_ingest_device(0, fmt, 2, 48000, FRAMES, src.data_ptr())
    ctx.sync()
    dt_ing = (time.perf_counter() - t0) / ITERS
    alg = 2 * FRAMES * 2 * eb
    out.append({"op": "ingest_device", "fmt": fmt, "frames": FRAMES, "channels": 2, "ms_wall": dt_ing * 1e3,
                "algorithmic_bytes": alg, "GBps_wall": alg / dt_ing / 1e9})
    for q in (0, 1):
        ctx.build_mipmaps(0, q)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(ITERS):
            ctx.build_mipmaps(0, q)
        ctx.sync()
        dt_mip = (time.perf_counter() - t0) / ITERS
        levels = ctx.L.wbx_mip_levels(FRAMES)
        wr = sum(ctx.L.wbx_mip_data_count(FRAMES, l) for l in range(levels)) * (2 if q else 1)
        alg = 2 * (FRAMES * eb + wr)
        out.append({"op": "build_mipmaps", "fmt": fmt, "quality": q, "frames": FRAMES, "channels": 2, "levels": levels,
                    "ms_wall": dt_mip * 1e3, "algorithmic_bytes": alg, "GBps_wall": alg / dt_mip / 1e9})
    del src
# host-side entry points (what the boundary does when it is handed host buffers): PCIe-inclusive rates
HF = 32 << 20
for fmt, npdt in (("i16", np.int16), ("f32", np.float32)):
    a = (np.arange(HF * 2, dtype=np.int64) % 30011).astype(npdt).reshape(HF, 2)
    pinned = torch.from_numpy(a).pin_memory()
    for name, arr in (("pageable", a), ("pinned", pinned.numpy())):
        ctx.clip_upload_interleaved(1, fmt, 48000, arr)
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.clip_upload_interleaved(1, fmt, 48000, arr)
        dt_up = (time.perf_counter() - t0) / 3
        out.append({"op": "upload_interleaved", "host_memory": name, "fmt": fmt, "frames": HF, "channels": 2,
                    "ms_wall": dt_up * 1e3, "host_bytes": arr.nbytes, "GBps_host_to_planar_hbm": arr.nbytes / dt_up / 1e9})
ctx.close()
for o in out:
    print(json.dumps(o))
