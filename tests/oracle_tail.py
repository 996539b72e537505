"""Oracle answers for blocks DEEP inside a long render (TEST INFRASTRUCTURE: only tests/ and bench.py's verify leg use it).

The CPU oracle is the reference's single thread: 4096 tracks cost it ~20 ms per block, so the head of a 2048-block render
is all a test could afford to check — and chained renders do their interesting work (chain words, epoch tags, the XCD-level
hand-over, the bounded sum grid) far behind the head.  This module gets the oracle to block 2047 in seconds without
changing one addition of the blocks that are compared:

  * tracks are independent (no sends / side chains in the reference), so the session is cut into contiguous SHARDS of
    tracks, one oracle engine each.  On the blocks that are not compared the shards simply advance, side by side on the
    host's cores (the sequencer and sampler state of a track depends on nothing but the track);
  * on a block that IS compared the shards run one after the other, each continuing the running un-clamped sum of the
    one before it (wbo_engine_process_from: Engine::process without the output clear) — addition for addition one
    Engine::process over all tracks, the last one clamps.  With sub-buses a shard holds whole buses, in order;
  * clip audio exists only where a compared block reads it: every sample array has its full length (calloc: untouched
    pages cost nothing) and is filled from the keyed generator in windows around the compared blocks.  A block that is
    not compared reads zeros — its output is never looked at, its sampler arithmetic (positions, lengths) does not depend
    on sample values.  A window that missed a read would show up as a mismatch, never as a false pass.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import math
import os
import threading
from typing import Dict, List, Sequence, Tuple

import numpy as np

import oracle_ffi as O
from whitebox_amd import synth


@dataclasses.dataclass
class TrackDesc:
    seed: int
    key_track: int                  # generator key (the global track index of bench sessions)
    fmt: str
    channels: int
    rate: int
    frames: int
    amp: float
    volume_db: float
    pan: float
    mute: bool
    bus: int                        # -1: straight into the master
    clips: List[Tuple[float, float, float, float, float]]   # (min_beat, max_beat, start_offset, speed, gain)


def descs_from_spec(spec) -> List[TrackDesc]:
    """a SessionSpec whose clips all play their own track's sample (the BASELINE configs, cut sessions)"""
    out = []
    for t in range(spec.n_tracks):
        s = spec.samples[t]
        out.append(TrackDesc(spec.seed, s.seed_track, s.fmt, s.channels, s.rate, s.frames, s.amp, spec.volumes_db[t], spec.pans[t],
                             spec.mutes[t], spec.track_bus[t] if spec.track_bus is not None else -1, []))
    for c in spec.clips:
        assert c.sample is None or c.sample == c.track
        out[c.track].clips.append((c.min_beat, c.max_beat, c.start_offset, c.speed, c.gain))
    return out


def _sparse_channel(d: TrackDesc, chan: int, windows: Sequence[Tuple[int, int]]) -> np.ndarray:
    dt = {"f32": np.float32, "i16": np.int16, "i24": np.int32, "i32": np.int32}[d.fmt]
    a = np.zeros(d.frames + 16, dtype=dt)
    for (lo, hi) in windows:
        lo, hi = max(0, lo), min(d.frames, hi)
        if hi <= lo:
            continue
        if d.fmt == "f32":
            a[lo:hi] = synth.clip_channel(d.seed, d.key_track, chan, hi - lo, d.amp, first=lo)
        elif d.fmt == "i16":
            a[lo:hi] = synth.clip_channel_i16(d.seed, d.key_track, chan, hi - lo, first=lo)
        else:
            a[lo:hi] = synth.clip_channel_i32(d.seed, d.key_track, chan, hi - lo, 24 if d.fmt == "i24" else 32, first=lo)
    return a


class ShardedOracle:
    def __init__(self, descs: List[TrackDesc], check_blocks: Sequence[int], *, block=512, channels=2, sample_rate=48000,
                 bpm=120.0, n_buses=0, threads=None, margin=96):
        self.check = sorted(set(int(b) for b in check_blocks))
        self.N, self.F, self.Cn = len(descs), block, channels
        beat_frames = sample_rate * 60.0 / bpm
        horizon = (self.check[-1] + 2) * block / beat_frames
        P = threads or max(1, min((os.cpu_count() or 2), 64, self.N // 8 or 1))
        # shard boundaries: contiguous track ranges; with sub-buses a shard holds whole buses (tracks of a bus are contiguous
        # in these sessions, bus ids ascend with the track index)
        if n_buses:
            assert all(d.bus >= 0 for d in descs) and all(descs[i].bus <= descs[i + 1].bus for i in range(self.N - 1))
            P = min(P, n_buses)
            cut_bus = [round(g * n_buses / P) for g in range(P + 1)]
            first_of_bus = {}
            for i, d in enumerate(descs):
                first_of_bus.setdefault(d.bus, i)
            cuts = [first_of_bus.get(b, self.N) if b < n_buses else self.N for b in cut_bus]
        else:
            cuts = [round(g * self.N / P) for g in range(P + 1)]
        self.shards = []
        self._keep = []
        for g in range(P):
            t0, t1 = cuts[g], cuts[g + 1]
            if t1 <= t0:
                continue
            e = O.OracleEngine(channels, block, sample_rate)
            e.set_bpm(bpm)
            bus0 = descs[t0].bus if n_buses else 0
            if n_buses:
                e.set_buses(descs[t1 - 1].bus - bus0 + 1)
            for i, t in enumerate(range(t0, t1)):
                d = descs[t]
                ps_rate = d.rate / sample_rate
                wins = []
                for (mn, mx, so, sp, _g) in d.clips:
                    c0, c1 = mn * beat_frames, mx * beat_frames
                    for b in self.check:
                        lo_f, hi_f = max(b * block, c0), min((b + 1) * block, c1)
                        if hi_f <= lo_f - 1:
                            continue
                        ps = ps_rate * sp
                        wins.append((int(so + (lo_f - c0) * ps) - margin, int(math.ceil(so + (hi_f - c0) * ps)) + margin))
                chans = [_sparse_channel(d, c, wins) for c in range(d.channels)]
                self._keep.append(chans)
                sid = e.add_sample(d.fmt, d.channels, d.rate, d.frames, chans)
                e.add_track()
                e.set_volume(i, d.volume_db)
                e.set_pan(i, d.pan)
                if d.mute:
                    e.set_mute(i, True)
                if n_buses:
                    e.set_bus(i, d.bus - bus0)
                for (mn, mx, so, sp, gn) in d.clips:
                    if mn <= horizon:
                        e.add_audio_clip(i, mn, mx, so, sid, sp, gn)
            self.shards.append((t0, t1, e))

    def close(self):
        for (_, _, e) in self.shards:
            e.close()
        self.shards = []

    def _advance(self, n: int):
        """every shard n blocks on, side by side (ctypes drops the GIL for the call)"""
        if n <= 0:
            return
        L = O.lib()

        def work(e):
            out = [np.zeros(self.F, np.float32) for _ in range(self.Cn)]
            ptrs = O.planar_ptrs(out)
            for _ in range(n):
                L.wbo_engine_process(e.e, ptrs, None)

        ths = [threading.Thread(target=work, args=(e,)) for (_, _, e) in self.shards]
        for th in ths:
            th.start()
        for th in ths:
            th.join()

    def run(self) -> Dict[int, Tuple[np.ndarray, np.ndarray, list]]:
        """-> {block: (master [C][F], peaks [N][2], stream-call rows (track, dst_start, len, offset bits, speed bits, gain bits))}"""
        res = {}
        for (_, _, e) in self.shards:
            e.enable_seglog()
            e.play()
        at = 0
        for b in self.check:
            self._advance(b - at)
            running = np.zeros((self.Cn, self.F), np.float32)
            peaks = np.zeros((self.N, 2), np.float32)
            rows = []
            for k, (t0, t1, e) in enumerate(self.shards):
                running = e.process_from(running, clamp=(k == len(self.shards) - 1))
                peaks[t0:t1] = e.peaks()
                rows += [(t + t0, ds, min(ln, 0xFFFF), O.f64_bits(off), O.f64_bits(spd), O.f32_bits(g))
                         for (t, ds, ln, off, spd, g, smp) in e.seglog()]
            res[b] = (running, peaks, rows)
            at = b + 1
        return res


def oracle_at_blocks(descs, check_blocks, **kw):
    so = ShardedOracle(descs, check_blocks, **kw)
    try:
        return so.run()
    finally:
        so.close()
