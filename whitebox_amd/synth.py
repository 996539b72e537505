"""Deterministic synthetic multitrack sessions (integer-hash generator).

Input generation only — no mixing arithmetic lives here.  The same generator exists as a HIP kernel
(`wbx_clip_synth`, csrc/wbx_kernels.hip) so large sessions never cross PCIe; both produce identical
bits: u = splitmix64(seed ^ (track<<40) ^ (chan<<32) ^ frame); v = ((u>>40) - 2^23) * 2^-23 in
[-1, 1); sample = fl32(v * amp).
"""
from __future__ import annotations

import dataclasses
import math
from typing import List, Optional, Tuple

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser over uint64 (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def clip_key(seed: int, track: int, chan: int) -> np.uint64:
    return np.uint64((seed ^ (track << 40) ^ (chan << 32)) & 0xFFFFFFFFFFFFFFFF)


def clip_channel(seed: int, track: int, chan: int, frames: int, amp: float, first: int = 0) -> np.ndarray:
    """fp32 samples [first, first+frames) of one clip channel."""
    idx = np.arange(first, first + frames, dtype=np.uint64)
    u = splitmix64(clip_key(seed, track, chan) ^ idx)
    q = (u >> np.uint64(40)).astype(np.int64) - (1 << 23)
    v = q.astype(np.float32) * np.float32(2.0 ** -23)      # exact: 24-bit integer * 2^-23
    return (v * np.float32(amp)).astype(np.float32)


def clip_channel_i16(seed: int, track: int, chan: int, frames: int, first: int = 0) -> np.ndarray:
    idx = np.arange(first, first + frames, dtype=np.uint64)
    u = splitmix64(clip_key(seed, track, chan) ^ idx)
    return ((u >> np.uint64(48)).astype(np.int64) - 32768).astype(np.int16)


def clip_channel_i32(seed: int, track: int, chan: int, frames: int, bits: int = 32, first: int = 0) -> np.ndarray:
    idx = np.arange(first, first + frames, dtype=np.uint64)
    u = splitmix64(clip_key(seed, track, chan) ^ idx)
    q = (u >> np.uint64(64 - bits)).astype(np.int64) - (1 << (bits - 1))
    return q.astype(np.int32)


def hash01(seed: int, track: int, salt: int) -> float:
    """h in [0,1) as a double: top 53 bits of the hash."""
    u = int(splitmix64(np.uint64((seed ^ (track << 40) ^ salt) & 0xFFFFFFFFFFFFFFFF)))
    return (u >> 11) * (2.0 ** -53)


def track_params(seed: int, track: int) -> Tuple[np.float32, np.float32]:
    """(volume_dB, pan) as the fp32 values handed to Track::set_volume / set_pan."""
    vol_db = np.float32(-12.0 + 18.0 * hash01(seed, track, 0xA1))
    pan = np.float32(2.0 * hash01(seed, track, 0xB2) - 1.0)
    return vol_db, pan


@dataclasses.dataclass
class ClipSpec:
    track: int
    min_beat: float
    max_beat: float
    start_offset: float = 0.0      # samples
    speed: float = 1.0
    gain: float = 1.0
    sample: Optional[int] = None   # index into SessionSpec.samples; None -> the track's own sample


@dataclasses.dataclass
class SampleSpec:
    seed_track: int                # generator key
    channels: int
    rate: int
    frames: int
    fmt: str = "f32"               # f32 | i16 | i24 | i32
    amp: float = 1.0


@dataclasses.dataclass
class SessionSpec:
    """A synthetic session, buildable through either engine's own API (oracle or product)."""
    name: str
    n_tracks: int
    seed: int
    samples: List[SampleSpec]
    clips: List[ClipSpec]
    volumes_db: List[float]
    pans: List[float]
    mutes: List[bool]
    n_buses: int = 0
    track_bus: Optional[List[int]] = None
    bpm: float = 120.0
    sample_rate: int = 48000
    block: int = 512
    channels: int = 2
    playhead_start: float = 0.0

    def sample_data(self, i: int) -> List[np.ndarray]:
        """Planar channel arrays of sample i, each with the reference's 16 zero frames of padding
        (dsp/sample.h:19, sample.cpp:127,140)."""
        s = self.samples[i]
        out = []
        for c in range(s.channels):
            if s.fmt == "f32":
                a = clip_channel(self.seed, s.seed_track, c, s.frames, s.amp)
            elif s.fmt == "i16":
                a = clip_channel_i16(self.seed, s.seed_track, c, s.frames)
            elif s.fmt == "i24":
                a = clip_channel_i32(self.seed, s.seed_track, c, s.frames, 24)
            elif s.fmt == "i32":
                a = clip_channel_i32(self.seed, s.seed_track, c, s.frames, 32)
            else:
                raise ValueError(s.fmt)
            out.append(np.concatenate([a, np.zeros(16, dtype=a.dtype)]))
        return out


def default_amp(n_tracks: int) -> float:
    """amp = 0.25/sqrt(N) as an fp32 value (master peak ~0.4: no clipping)."""
    return float(np.float32(0.25 / math.sqrt(n_tracks)))


def make_session(name: str, n_tracks: int, *, clip_channels: int = 2, src_rate: int = 48000, n_blocks: int = 8,
                 n_buses: int = 0, seed: int = 0x5EED0000, unity_gain: bool = False, amp: Optional[float] = None,
                 seek: bool = False, block: int = 512, sample_rate: int = 48000, bpm: float = 120.0,
                 fmt: str = "f32") -> SessionSpec:
    """The BASELINE.json configs as SessionSpecs.

    One clip per track starting at beat 0, long enough for n_blocks (+2) blocks at the clip's rate.
    `seek=True` adds the mid-block start/stop variant (clips that begin and end inside blocks and a
    second clip on every 4th track).
    """
    if amp is None:
        amp = default_amp(n_tracks)
    beat_frames = sample_rate * 60.0 / bpm
    need = int(math.ceil((n_blocks + 2) * block * (src_rate / sample_rate))) + 32
    samples, clips, vols, pans, mutes = [], [], [], [], []
    for t in range(n_tracks):
        samples.append(SampleSpec(seed_track=t, channels=clip_channels, rate=src_rate, frames=need, fmt=fmt,
                                  amp=amp if fmt == "f32" else 1.0))
        if unity_gain:
            vols.append(0.0)
            pans.append(0.0)
        else:
            v, p = track_params(seed, t)
            vols.append(float(v))
            pans.append(float(p))
        mutes.append(False)
        if not seek:
            clips.append(ClipSpec(t, 0.0, (n_blocks + 1) * block / beat_frames))
        else:
            # starts at frame 100+7t%300 of block 0, ends inside block n_blocks-2; every 4th track has a
            # second clip starting a little later in the block where the first one ended
            s0 = 100 + (7 * t) % 300
            e0 = (n_blocks - 2) * block + 50 + (11 * t) % 200
            clips.append(ClipSpec(t, s0 / beat_frames, e0 / beat_frames, start_offset=float(10 + t % 5),
                                  gain=0.5 if t % 3 == 0 else 1.0))
            if t % 4 == 0:
                s1 = e0 + 33 + t % 17
                clips.append(ClipSpec(t, s1 / beat_frames, (n_blocks + 1) * block / beat_frames,
                                      start_offset=0.0))
    track_bus = None
    if n_buses:
        per = max(1, n_tracks // n_buses)
        track_bus = [min(t // per, n_buses - 1) for t in range(n_tracks)]
    return SessionSpec(name=name, n_tracks=n_tracks, seed=seed, samples=samples, clips=clips, volumes_db=vols,
                       pans=pans, mutes=mutes, n_buses=n_buses, track_bus=track_bus, bpm=bpm,
                       sample_rate=sample_rate, block=block)


def cut_into_clips(spec: SessionSpec, clip_blocks: float, session_blocks: int) -> SessionSpec:
    """Every track of a one-clip-per-track session cut into back-to-back clips of `clip_blocks` blocks, each reading on
    from where the previous one stopped, staggered per track (bench.py --clip-blocks builds the same layout)."""
    beat_frames = spec.sample_rate * 60.0 / spec.bpm
    L = clip_blocks * spec.block
    clips = []
    for t in range(spec.n_tracks):
        rate = spec.samples[t].rate
        pos = -((t * 37) % 512) / 512.0 * L
        while pos < (session_blocks + 1) * spec.block:
            a, b = max(pos, 0.0), pos + L
            clips.append(ClipSpec(t, a / beat_frames, b / beat_frames, start_offset=a * (rate / spec.sample_rate)))
            pos = b
    return dataclasses.replace(spec, clips=clips)
