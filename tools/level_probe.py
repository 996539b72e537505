#!/usr/bin/env python3
"""tools/level_probe.py — where the grouped summation order stands against the 1e-6 RMS budget as the session gets hotter.

The reference adds tracks strictly sequentially (engine.cpp:1600-1617).  Renders shorter than 1024 blocks — and the
one-block audio callback — add 128-track (64 / 32 / 16-track) groups in order and then the group sums: per-track values are
identical, only the association of the fp32 additions differs, and that error scales with the level of the running sum.
This prints, for N = 4096 (c3: 44.1 kHz clips; c4: 64 buses) and the 8-way sharded N = 32768 (c5), RMS and max-abs of
(device master - oracle master) at session levels amp = m / sqrt(N), m = 0.25 (the synthetic default) ... 4, with the
share of samples the master clamp holds at +-1 (those carry no error).  Output is committed under profiles/ and quoted
by DESIGN.md "Summation order".   python tools/level_probe.py [--blocks 4] > profiles/r03_level_probe.txt
"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402

import oracle_ffi as O  # noqa: E402
from whitebox_amd import synth  # noqa: E402
from whitebox_amd.engine import build_engine  # noqa: E402


def stats(m, om):
    d = m.astype(np.float64) - om.astype(np.float64)
    un = np.abs(om) < 1.0
    return (float(np.sqrt(np.mean(d * d))), float(np.abs(d).max()), float(1.0 - un.mean()),
            float(np.sqrt(np.mean(d[un] ** 2))) if un.any() else 0.0, float(np.sqrt(np.mean(om.astype(np.float64) ** 2))))


def oracle_master(spec, K, clamp=True):
    e = O.build_oracle_engine(spec)
    e.play()
    out = np.stack([e.process(clamp=clamp)[0] for _ in range(K)])
    e.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=4)
    ap.add_argument("--mults", default="0.25,0.5,1,2,4")
    a = ap.parse_args()
    K = a.blocks
    mults = [float(x) for x in a.mults.split(",")]
    print("workload  N      groups            amp*sqrt(N)  master_rms  clamped  rms(all)    rms(unclamped)  max_abs")
    for name, kw in (("c3", dict(src_rate=44100)), ("u4096", dict()), ("c4", dict(n_buses=64))):
        N = 4096
        for m in mults:
            amp = float(np.float32(m / math.sqrt(N)))
            spec = synth.make_session(name, N, n_blocks=K, seed=0x5EED0003, amp=amp, **kw)
            om = oracle_master(spec, K)
            for label, gs, mb in (("128 (render-ahead)", 0, K), ("16 (callback)", 0, 1), ("whole list", N, K)):
                if label != "128 (render-ahead)" and name == "c4":
                    continue
                eng = build_engine(spec, max_blocks=mb, group_size=gs)
                eng.play()
                if mb == 1:
                    from whitebox_amd import AudioBuffer
                    out = AudioBuffer(spec.block, 2)
                    got = []
                    for _ in range(K):
                        eng.process(None, out, 48000.0)
                        got.append(np.stack(out.channel_buffers).copy())
                    got = np.stack(got)
                else:
                    eng.render(K)
                    got, _, _ = eng.ctx.fetch()
                eng.close()
                r, mx, cl, ru, mr = stats(got, om)
                print(f"{name:9s} {N:<6d} {label:18s} {m:<11g}  {mr:<10.3f}  {cl:<7.3f}  {r:<10.3e}  {ru:<14.3e}  {mx:.3e}", flush=True)
    # c5: 8 shards of 4096 tracks, each summed in 128-track groups, partial masters added in rank order, clamp after the sum
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_dist_gloo import _shard_spec
    from whitebox_amd.dist import shard_tracks
    N, world, K5 = 32768, 8, 2
    for m in (0.25, 1.0, 2.0):
        amp = float(np.float32(m / math.sqrt(N)))
        spec = synth.make_session("c5", N, n_blocks=K5, seed=0x5EED0006, amp=amp)
        om = oracle_master(spec, K5)
        for label, gs in (("8 x 128-groups", 0), ("8 x whole shard", 4096)):
            total = np.zeros_like(om)
            for rank in range(world):
                first, count = shard_tracks(N, world, rank)
                eng = build_engine(_shard_spec(spec, first, count), max_blocks=K5, group_size=gs)
                eng.ctx.set_clamp(False)
                eng.play()
                eng.render(K5)
                part, _, _ = eng.ctx.fetch()
                total = (total + part).astype(np.float32)
                eng.close()
            got = np.clip(total, -1.0, 1.0).astype(np.float32)
            r, mx, cl, ru, mr = stats(got, om)
            print(f"{'c5':9s} {N:<6d} {label:18s} {m:<11g}  {mr:<10.3f}  {cl:<7.3f}  {r:<10.3e}  {ru:<14.3e}  {mx:.3e}", flush=True)


if __name__ == "__main__":
    main()
