"""ctypes driver of tests/cpp/host_sim.cpp: the product's layer-2 host code (wbx_host.h + the sequencer source of
plan_kernel, wbx_seq.h) compiled with plain g++ and run on the CPU.  TEST INFRASTRUCTURE — nothing here is shipped;
it gives the seek math, the clip edits and the two-thread contract coverage that needs no GPU."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List

import numpy as np

from whitebox_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_sim.cpp")
HDRS = [os.path.join(ROOT, "whitebox_amd", "csrc", h) for h in ("wbx_host.h", "wbx_seq.h", "wbx_clip_edit.h", "wbx_dev.h")]
BUILD = os.path.join(ROOT, "tests", "cpp", "_build")
# no FMA contraction (the reference build has none), and the records are viewed as 16-B quads
FLAGS = ["-std=c++20", "-ffp-contract=off", "-fno-fast-math", "-fno-strict-aliasing", "-Wall"]

_lib = None


def _stale(out: str) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(p) > t for p in [SRC] + HDRS)


def build_lib() -> str:
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, "libwbxhostsim.so")
    if _stale(out):   # (to a name of its own, then renamed: pytest-xdist workers build side by side and must never load half a file)
        tmp = f"{out}.{os.getpid()}.tmp"
        subprocess.check_call(["g++", *FLAGS, "-O2", "-shared", "-fPIC", SRC, "-o", tmp])
        os.replace(tmp, out)
    return out


def build_tsan() -> str:
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, "host_tsan")
    if _stale(out):
        tmp = f"{out}.{os.getpid()}.tmp"
        subprocess.check_call(["g++", *FLAGS, "-O1", "-g", "-fsanitize=thread", "-DHOST_SIM_MAIN", SRC, "-o", tmp, "-lpthread"])
        os.replace(tmp, out)
    return out


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        L = C.CDLL(build_lib())
        L.hsim_create.restype = C.c_void_p
        L.hsim_create.argtypes = [C.c_uint32] * 5
        L.hsim_destroy.argtypes = [C.c_void_p]
        L.hsim_template_capacity.restype = C.c_uint32
        L.hsim_template_capacity.argtypes = [C.c_void_p]
        L.hsim_set_masked_rows.restype = None
        L.hsim_set_masked_rows.argtypes = [C.c_void_p, C.c_uint32]
        L.hsim_set_segments.restype = None
        L.hsim_set_segments.argtypes = [C.c_void_p, C.c_uint32]
        L.hsim_segment_stats.restype = None
        L.hsim_segment_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        sig = {
            "hsim_set_bpm": [C.c_double], "hsim_set_playhead_position": [C.c_double],
            "hsim_add_track": [C.POINTER(C.c_uint32)],
            "hsim_track_set_volume": [C.c_uint32, C.c_float], "hsim_track_set_pan": [C.c_uint32, C.c_float],
            "hsim_track_set_mute": [C.c_uint32, C.c_int], "hsim_solo_track": [C.c_uint32],
            "hsim_delete_track": [C.c_uint32], "hsim_move_track": [C.c_uint32, C.c_uint32],
            "hsim_add_sample": [C.c_int, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32)],
            "hsim_add_audio_clip": [C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_double, C.c_float],
            "hsim_move_clip": [C.c_uint32, C.c_uint32, C.c_double],
            "hsim_resize_clip": [C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int],
            "hsim_delete_clip": [C.c_uint32, C.c_uint32], "hsim_delete_region": [C.c_uint32, C.c_double, C.c_double],
            "hsim_set_clip_gain": [C.c_uint32, C.c_uint32, C.c_float],
            "hsim_clip_count": [C.c_uint32, C.POINTER(C.c_uint32)],
            "hsim_get_clip": [C.c_uint32, C.c_uint32, C.POINTER(_ffi.ClipInfo)],
            "hsim_play": [], "hsim_stop": [], "hsim_render": [C.c_uint32],
            "hsim_plan_counters": [C.POINTER(C.c_uint32)],
            "hsim_row_kinds": [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_size_t],
            "hsim_fetch_plan": [C.POINTER(_ffi.PlanRecord), C.c_size_t, C.POINTER(C.c_size_t)],
            "hsim_gains": [C.POINTER(C.c_float), C.c_uint32],
            "hsim_transport": [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)],
            "hsim_thread_stats": [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint32],
        }
        for name, args in sig.items():
            fn = getattr(L, name)
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p] + args
        _lib = L
    return _lib


class SimTrack:
    def __init__(self, eng: "HostSimEngine", index: int):
        self.engine, self.index = eng, index

    def set_volume(self, db): self.engine._ok(self.engine.L.hsim_track_set_volume(self.engine.h, self.index, np.float32(db)))
    def set_pan(self, p): self.engine._ok(self.engine.L.hsim_track_set_pan(self.engine.h, self.index, np.float32(p)))
    def set_mute(self, m): self.engine._ok(self.engine.L.hsim_track_set_mute(self.engine.h, self.index, int(m)))


class HostSimEngine:
    """The method names of whitebox_amd.engine.Engine (the reference's), over the CPU harness."""

    def __init__(self, max_tracks, buffer_size=512, sample_rate=48000, output_channels=2, max_blocks=1):
        self.L = lib()
        self.h = C.c_void_p(self.L.hsim_create(max_tracks, max_blocks, buffer_size, output_channels, sample_rate))
        self.tracks: List[SimTrack] = []

    def _ok(self, st):
        assert st == 0, f"host sim call failed: {st}"

    def close(self):
        if self.h:
            self.L.hsim_destroy(self.h)
        self.h = None

    def set_bpm(self, bpm): self._ok(self.L.hsim_set_bpm(self.h, bpm))
    def set_playhead_position(self, beat): self._ok(self.L.hsim_set_playhead_position(self.h, beat))

    def add_track(self, name=""):
        i = C.c_uint32()
        self._ok(self.L.hsim_add_track(self.h, C.byref(i)))
        t = SimTrack(self, i.value)
        self.tracks.append(t)
        return t

    def _reindex(self):
        for i, t in enumerate(self.tracks):
            t.index = i

    def delete_track(self, slot):
        self._ok(self.L.hsim_delete_track(self.h, slot))
        del self.tracks[slot]
        self._reindex()

    def move_track(self, a, b):
        self._ok(self.L.hsim_move_track(self.h, a, b))
        t = self.tracks.pop(a)
        self.tracks.insert(b, t)
        self._reindex()

    def solo_track(self, slot): self._ok(self.L.hsim_solo_track(self.h, slot))

    def add_sample_meta(self, fmt, channels, rate, frames) -> int:
        i = C.c_uint32()
        self._ok(self.L.hsim_add_sample(self.h, _ffi.FMT[fmt], channels, rate, frames, C.byref(i)))
        return i.value

    def add_audio_clip(self, track, name, mn, mx, so, sample, speed=1.0, gain=1.0):
        self._ok(self.L.hsim_add_audio_clip(self.h, track.index, mn, mx, so, sample, speed, np.float32(gain)))

    def move_clip(self, track, clip, rel): self._ok(self.L.hsim_move_clip(self.h, track.index, clip, rel))

    def resize_clip(self, track, clip, rel, limit, min_length, left, shift=False, stretch=False):
        self._ok(self.L.hsim_resize_clip(self.h, track.index, clip, rel, limit, min_length, int(left), int(shift), int(stretch)))

    def delete_clip(self, track, clip): self._ok(self.L.hsim_delete_clip(self.h, track.index, clip))
    def delete_region(self, track, mn, mx): self._ok(self.L.hsim_delete_region(self.h, track.index, mn, mx))
    def set_clip_gain(self, track, clip, g): self._ok(self.L.hsim_set_clip_gain(self.h, track.index, clip, np.float32(g)))

    def clips(self, track):
        n = C.c_uint32()
        self._ok(self.L.hsim_clip_count(self.h, track.index, C.byref(n)))
        out = []
        for i in range(n.value):
            ci = _ffi.ClipInfo()
            self._ok(self.L.hsim_get_clip(self.h, track.index, i, C.byref(ci)))
            out.append((ci.min_time, ci.max_time, ci.start_offset, ci.speed, ci.gain, ci.sample))
        return out

    def play(self): self._ok(self.L.hsim_play(self.h))
    def stop(self): self._ok(self.L.hsim_stop(self.h))
    def render(self, k): self._ok(self.L.hsim_render(self.h, k))

    def plan_counters(self):
        a = (C.c_uint32 * 4)()
        self._ok(self.L.hsim_plan_counters(self.h, a))
        return list(a)

    def template_capacity(self): return self.L.hsim_template_capacity(self.h)

    def set_segments(self, seg_len):
        """plan batch renders by segments of seg_len blocks (wbx_seq.h plan_segment), 0 = one walk per track"""
        self.L.hsim_set_segments(self.h, int(seg_len))

    def segment_stats(self):
        """(renders planned by segments, tracks with a seam that did not hold, segments planned again, speculative lanes)"""
        out = (C.c_uint64 * 4)()
        self.L.hsim_segment_stats(self.h, out)
        return tuple(int(x) for x in out)

    def set_masked_rows(self, level):
        """plan as for a mix instance that renders partial-coverage rows / ROW_PAIRs in its hot loop (PlanArgs::masked_rows:
        1 / True fp32 rows, 2 also integer PCM at unity speed)"""
        self.L.hsim_set_masked_rows(self.h, int(level))

    def row_kinds(self, n_rows: int):
        """(row flags, template kind) per (block, track) of the last render"""
        fl, kd = (C.c_uint32 * n_rows)(), (C.c_uint32 * n_rows)()
        self._ok(self.L.hsim_row_kinds(self.h, fl, kd, n_rows))
        return list(fl), list(kd)

    def fetch_plan(self):
        n = C.c_size_t()
        self.L.hsim_fetch_plan(self.h, None, 0, C.byref(n))
        arr = (_ffi.PlanRecord * max(1, n.value))()
        self._ok(self.L.hsim_fetch_plan(self.h, arr, n.value, C.byref(n)))
        return [(r.block, r.track, r.buffer_offset, r.num_samples, r.num_actual, r.sample, r.sample_offset,
                 r.playback_speed, r.gain, r.flags) for r in arr[:n.value]]

    def gains(self):
        n = len(self.tracks)
        g = np.zeros((n, 2), np.float32)
        self._ok(self.L.hsim_gains(self.h, g.ctypes.data_as(C.POINTER(C.c_float)), n))
        return g

    def transport(self):
        ph, sp, pl = C.c_double(), C.c_double(), C.c_int()
        self._ok(self.L.hsim_transport(self.h, C.byref(ph), C.byref(sp), C.byref(pl)))
        return ph.value, sp.value, bool(pl.value)

    def thread_stats(self):
        n = len(self.tracks)
        seen = C.c_uint64()
        dr = (C.c_uint64 * max(1, n))()
        self._ok(self.L.hsim_thread_stats(self.h, C.byref(seen), dr, n))
        return seen.value, list(dr[:n])


def build_sim_engine(spec, max_blocks=8, masked_rows=False, segments=0) -> HostSimEngine:
    """The same construction sequence as whitebox_amd.engine.build_engine, without audio."""
    eng = HostSimEngine(max(spec.n_tracks, 1), spec.block, spec.sample_rate, spec.channels, max_blocks=max_blocks)
    eng.set_masked_rows(masked_rows)
    eng.set_segments(segments)
    eng.set_bpm(spec.bpm)
    if spec.playhead_start:
        eng.set_playhead_position(spec.playhead_start)
    ids = [eng.add_sample_meta(s.fmt, s.channels, s.rate, s.frames) for s in spec.samples]
    for t in range(spec.n_tracks):
        tr = eng.add_track(f"t{t}")
        tr.set_volume(spec.volumes_db[t])
        tr.set_pan(spec.pans[t])
        if spec.mutes[t]:
            tr.set_mute(True)
    for c in spec.clips:
        sidx = c.sample if c.sample is not None else c.track
        eng.add_audio_clip(eng.tracks[c.track], "clip", c.min_beat, c.max_beat, c.start_offset, ids[sidx], c.speed, c.gain)
    return eng
