"""ctypes bindings to the CPU oracle (oracle/liboracle.so) and, where it was built, to the reference's
own translation units (oracle/_ref/libwbref.so).  TEST INFRASTRUCTURE: imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")
REF_PATH = os.path.join(ORACLE_DIR, "_ref", "libwbref.so")

FMT = {"f32": 9, "i16": 3, "i24": 5, "i32": 7}
EV_NONE, EV_STOP, EV_PLAY = 0, 1, 2

c_f32p = C.POINTER(C.c_float)
c_f32pp = C.POINTER(c_f32p)
c_voidpp = C.POINTER(C.c_void_p)


def build_oracle(force: bool = False) -> None:
    src = os.path.join(ORACLE_DIR, "wb_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)


def build_ref() -> bool:
    """Build oracle/_ref from /root/reference when it exists (this container only)."""
    if not os.path.isdir("/root/reference/src"):
        return os.path.exists(REF_PATH)
    subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return True


class _Event(C.Structure):
    _fields_ = [("type", C.c_int), ("buffer_offset", C.c_uint32), ("time", C.c_double), ("speed", C.c_double),
                ("sample_offset", C.c_uint64), ("clip", C.c_int)]


class _Sampler(C.Structure):
    _fields_ = [("playback_speed", C.c_double), ("sample_offset", C.c_double)]


class _Msg(C.Structure):
    _fields_ = [("id", C.c_uint32), ("value", C.c_double)]


class _Track(C.Structure):
    _fields_ = [("clips", C.c_void_p), ("n_clips", C.c_uint32), ("cap_clips", C.c_uint32),
                ("has_clip_idx", C.c_int), ("clip_idx", C.c_uint32), ("refresh_voice", C.c_int),
                ("partially_ended", C.c_int),
                ("events", _Event * 64), ("n_events", C.c_uint32),
                ("current_event", _Event), ("cur_gain", C.c_float), ("cur_sample", C.c_int), ("cur_clip_uid", C.c_uint32),
                ("sampler", _Sampler),
                ("volume", C.c_float), ("pan", C.c_float), ("pan_coeffs", C.c_float * 2), ("mute", C.c_int),
                ("msgs", _Msg * 64), ("n_msgs", C.c_uint32),
                ("level", C.c_float * 2), ("block_peak", C.c_float * 2), ("bus", C.c_int), ("ui_solo", C.c_int),
                ("free_uids", C.c_void_p), ("n_free_uids", C.c_uint32), ("cap_free_uids", C.c_uint32)]


class Clip(C.Structure):
    _fields_ = [("min_time", C.c_double), ("max_time", C.c_double), ("start_offset", C.c_double), ("speed", C.c_double),
                ("gain", C.c_float), ("sample", C.c_int), ("internal_state_changed", C.c_int), ("deleted", C.c_int),
                ("uid", C.c_uint32)]


class SegLog(C.Structure):
    _fields_ = [("playback_speed", C.c_double), ("sample_offset", C.c_double), ("track", C.c_uint32),
                ("dst_start", C.c_uint32), ("len", C.c_uint32), ("gain", C.c_float), ("sample", C.c_int)]


class _Engine(C.Structure):
    _fields_ = [("tracks", C.POINTER(_Track)), ("n_tracks", C.c_uint32), ("cap_tracks", C.c_uint32),
                ("samples", C.c_void_p), ("n_samples_tab", C.c_uint32), ("cap_samples", C.c_uint32),
                ("out_channels", C.c_uint32), ("buffer_size", C.c_uint32), ("sample_rate", C.c_uint32),
                ("ppq", C.c_double), ("playhead", C.c_double), ("playhead_start", C.c_double),
                ("sample_position", C.c_double), ("beat_duration", C.c_double), ("playing", C.c_int),
                ("n_buses", C.c_uint32), ("mixbuf", c_f32p * 16), ("busbuf", c_f32p),
                ("seglog", C.POINTER(SegLog)), ("n_seglog", C.c_uint32), ("cap_seglog", C.c_uint32),
                ("seglog_enabled", C.c_int), ("next_clip_uid", C.c_uint32)]


_lib = None
_ref = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(LIB_PATH)
        L.wbo_db_to_linear.restype = C.c_float
        L.wbo_db_to_linear.argtypes = [C.c_float]
        L.wbo_pan_coefs.argtypes = [C.c_float, C.c_int, c_f32p, c_f32p]
        L.wbo_beat_to_samples.restype = C.c_double
        L.wbo_beat_to_samples.argtypes = [C.c_double] * 3
        L.wbo_samples_to_beat.restype = C.c_double
        L.wbo_samples_to_beat.argtypes = [C.c_double] * 3
        L.wbo_perf_update.restype = C.c_double
        L.wbo_perf_update.argtypes = [C.c_double] * 3
        L.wbo_perf_get_usage.restype = C.c_double
        L.wbo_perf_get_usage.argtypes = [C.c_double]
        L.wbo_buffer_duration_ms.restype = C.c_double
        L.wbo_buffer_duration_ms.argtypes = [C.c_uint32, C.c_uint32]
        L.wbo_lower_bound_max_time.restype = C.c_uint32
        L.wbo_lower_bound_max_time.argtypes = [C.POINTER(C.c_double), C.c_uint32, C.c_double]
        L.wbo_deinterleave.restype = C.c_size_t
        L.wbo_deinterleave.argtypes = [c_voidpp, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_size_t]
        L.wbo_mip_levels.restype = C.c_uint32
        L.wbo_mip_levels.argtypes = [C.c_size_t]
        L.wbo_mip_data_count.restype = C.c_size_t
        L.wbo_mip_data_count.argtypes = [C.c_size_t, C.c_uint32]
        L.wbo_mip_summarize.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
        L.wbo_abs_max.restype = C.c_float
        L.wbo_abs_max.argtypes = [c_f32p, C.c_uint32]
        L.wbo_apply_gain.argtypes = [c_f32p, C.c_uint32, C.c_float]
        L.wbo_sampler_reset.argtypes = [C.POINTER(_Sampler), C.c_double, C.c_double, C.c_double, C.c_double]
        L.wbo_engine_create.restype = C.POINTER(_Engine)
        L.wbo_engine_create.argtypes = [C.c_uint32] * 3
        L.wbo_engine_destroy.argtypes = [C.POINTER(_Engine)]
        L.wbo_engine_set_bpm.argtypes = [C.POINTER(_Engine), C.c_double]
        L.wbo_engine_set_playhead.argtypes = [C.POINTER(_Engine), C.c_double]
        L.wbo_engine_set_buses.argtypes = [C.POINTER(_Engine), C.c_uint32]
        L.wbo_engine_add_sample.restype = C.c_int
        L.wbo_engine_add_sample.argtypes = [C.POINTER(_Engine), C.c_int, C.c_uint32, C.c_uint32, C.c_size_t, c_voidpp]
        L.wbo_engine_add_track.restype = C.c_int
        L.wbo_engine_add_track.argtypes = [C.POINTER(_Engine)]
        L.wbo_track_set_volume.argtypes = [C.POINTER(_Engine), C.c_int, C.c_float]
        L.wbo_track_set_pan.argtypes = [C.POINTER(_Engine), C.c_int, C.c_float]
        L.wbo_track_set_mute.argtypes = [C.POINTER(_Engine), C.c_int, C.c_int]
        L.wbo_track_set_bus.argtypes = [C.POINTER(_Engine), C.c_int, C.c_int]
        L.wbo_engine_add_audio_clip.restype = C.c_int
        L.wbo_engine_add_audio_clip.argtypes = [C.POINTER(_Engine), C.c_int, C.c_double, C.c_double, C.c_double,
                                                C.c_int, C.c_double, C.c_float]
        dp = C.POINTER(C.c_double)
        L.wbo_calc_move_clip.argtypes = [C.c_double] * 4 + [dp, dp]
        L.wbo_calc_resize_clip.argtypes = [C.c_double] * 11 + [C.c_int] * 4 + [dp] * 4
        L.wbo_calc_clip_shift.restype = C.c_double
        L.wbo_calc_clip_shift.argtypes = [C.c_double] * 4
        L.wbo_shift_clip_content.restype = C.c_double
        L.wbo_shift_clip_content.argtypes = [C.c_double] * 5
        L.wbo_engine_move_clip.argtypes = [C.POINTER(_Engine), C.c_int, C.c_uint32, C.c_double]
        L.wbo_engine_resize_clip.argtypes = [C.POINTER(_Engine), C.c_int, C.c_uint32, C.c_double, C.c_double, C.c_double,
                                             C.c_int, C.c_int, C.c_int]
        L.wbo_engine_delete_clip.argtypes = [C.POINTER(_Engine), C.c_int, C.c_uint32]
        L.wbo_engine_set_clip_gain.argtypes = [C.POINTER(_Engine), C.c_int, C.c_uint32, C.c_float]
        L.wbo_engine_delete_region.argtypes = [C.POINTER(_Engine), C.c_int, C.c_double, C.c_double]
        L.wbo_track_clip_count.restype = C.c_uint32
        L.wbo_track_clip_count.argtypes = [C.POINTER(_Engine), C.c_int]
        L.wbo_track_clip.restype = C.POINTER(Clip)
        L.wbo_track_clip.argtypes = [C.POINTER(_Engine), C.c_int, C.c_uint32]
        L.wbo_engine_play.argtypes = [C.POINTER(_Engine)]
        L.wbo_engine_stop.argtypes = [C.POINTER(_Engine)]
        L.wbo_engine_process.argtypes = [C.POINTER(_Engine), c_f32pp, c_f32p]
        L.wbo_engine_enable_seglog.argtypes = [C.POINTER(_Engine), C.c_int]
        L.wbo_engine_process_ex.argtypes = [C.POINTER(_Engine), c_f32pp, c_f32p, C.c_int]
        L.wbo_master_clamp.argtypes = [c_f32pp, C.c_uint32, C.c_uint32]
        for name, t in (("wbo_f32_to_interleaved_i16", C.c_void_p), ("wbo_f32_to_interleaved_i24", C.c_void_p),
                        ("wbo_f32_to_interleaved_i24_x8", C.c_void_p), ("wbo_f32_to_interleaved_i32", C.c_void_p),
                        ("wbo_f32_to_interleaved_f32", C.c_void_p)):
            getattr(L, name).argtypes = [t, c_f32pp, C.c_size_t, C.c_size_t, C.c_uint32]
        _lib = L
    return _lib


class RefSegment(C.Structure):
    _fields_ = [("playback_speed", C.c_double), ("sample_offset", C.c_double), ("track", C.c_uint32),
                ("dst_start", C.c_uint32), ("len", C.c_uint32), ("gain", C.c_float), ("sample", C.c_int)]


def ref() -> Optional[C.CDLL]:
    """The reference's own TUs (None where oracle/_ref was never built)."""
    global _ref
    if _ref is None:
        if not build_ref() or not os.path.exists(REF_PATH):
            return None
        R = C.CDLL(REF_PATH)
        R.ref_db_to_linear.restype = C.c_float
        R.ref_db_to_linear.argtypes = [C.c_float]
        R.ref_pan_coefs.argtypes = [C.c_float, C.c_int, c_f32p, c_f32p]
        R.ref_beat_to_samples.restype = C.c_double
        R.ref_beat_to_samples.argtypes = [C.c_double] * 3
        R.ref_samples_to_beat.restype = C.c_double
        R.ref_samples_to_beat.argtypes = [C.c_double] * 3
        R.ref_perf_update.restype = C.c_double
        R.ref_perf_update.argtypes = [C.c_double] * 3
        R.ref_perf_get_usage.restype = C.c_double
        R.ref_perf_get_usage.argtypes = [C.c_double]
        R.ref_buffer_duration_ms.restype = C.c_double
        R.ref_buffer_duration_ms.argtypes = [C.c_uint32, C.c_uint32]
        R.ref_apply_gain.argtypes = [c_f32p, C.c_uint32, C.c_float]
        R.ref_find_abs_maximum.restype = C.c_float
        R.ref_find_abs_maximum.argtypes = [c_f32p, C.c_uint32]
        R.ref_sampler_reset.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)] + [C.c_double] * 4
        R.ref_sampler_stream.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_uint32, C.c_uint32,
                                         C.c_size_t, c_voidpp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, c_f32pp]
        R.ref_mix_block.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(RefSegment), C.c_uint32,
                                    C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                    C.POINTER(C.c_size_t), C.POINTER(c_voidpp), c_f32p, C.POINTER(C.c_int), C.c_uint32,
                                    c_f32pp, c_f32p, c_f32p, C.POINTER(C.c_double), C.c_int]
        dp = C.POINTER(C.c_double)
        R.ref_calc_move_clip.argtypes = [C.c_double] * 4 + [dp, dp]
        R.ref_calc_resize_clip.argtypes = [C.c_double] * 11 + [C.c_int] * 4 + [dp] * 4
        R.ref_calc_clip_shift.restype = C.c_double
        R.ref_calc_clip_shift.argtypes = [C.c_double] * 4
        R.ref_shift_clip_content.restype = C.c_double
        R.ref_shift_clip_content.argtypes = [C.c_double] * 5
        R.ref_find_lower_bound_max_time.restype = C.c_uint32
        R.ref_find_lower_bound_max_time.argtypes = [C.POINTER(C.c_double), C.c_uint32, C.c_double]
        for name in ("ref_f32_to_i16", "ref_f32_to_i24", "ref_f32_to_i24_x8", "ref_f32_to_i32", "ref_f32_to_f32"):
            getattr(R, name).argtypes = [C.c_void_p, c_f32pp, C.c_size_t, C.c_size_t, C.c_uint32]
        _ref = R
    return _ref


_ref_mip = None
REF_MIP_PATH = os.path.join(ORACLE_DIR, "_ref", "libwbref_mip.so")


def ref_mip() -> Optional[C.CDLL]:
    """The reference's own summarize_for_mipmaps_impl (oracle/Makefile: cut out of gfx/waveform_visual.cpp where it lies and
    compiled unmodified; None where oracle/_ref was never built)."""
    global _ref_mip
    if _ref_mip is None:
        if not build_ref() or not os.path.exists(REF_MIP_PATH):
            return None
        R = C.CDLL(REF_MIP_PATH)
        R.ref_mip_summarize.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
        _ref_mip = R
    return _ref_mip


def ref_mip_level(fmt: str, data: np.ndarray, level: int, quality: int) -> np.ndarray:
    n = lib().wbo_mip_data_count(len(data), level)
    out = np.zeros(n, dtype=np.int16 if quality else np.int8)
    d = np.ascontiguousarray(data)
    ref_mip().ref_mip_summarize(FMT[fmt], len(d), d.ctypes.data, level, 16 if quality else 8, out.ctypes.data)
    return out


_ref_deint = None
REF_DEINT_PATH = os.path.join(ORACLE_DIR, "_ref", "libwbref_deint.so")


def ref_deint() -> Optional[C.CDLL]:
    """The reference's own deinterleave_samples<T> (oracle/Makefile: cut out of dsp/sample.cpp where it lies and compiled
    unmodified for any integer count type; None where oracle/_ref was never built)."""
    global _ref_deint
    if _ref_deint is None:
        if not build_ref() or not os.path.exists(REF_DEINT_PATH):
            return None
        R = C.CDLL(REF_DEINT_PATH)
        R.ref_deinterleave.restype = C.c_int64
        R.ref_deinterleave.argtypes = [c_voidpp, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int]
        R.ref_deinterleave_f32.restype = C.c_int64
        R.ref_deinterleave_f32.argtypes = [c_voidpp, C.c_void_p, C.c_int64, C.c_int, C.c_int64]
        _ref_deint = R
    return _ref_deint


def ref_deinterleave(interleaved: np.ndarray, chunk: int = 1024, count_bits: int = 64, pad: int = 16) -> List[np.ndarray]:
    """Sample::load_file's loop (sample.cpp:127-142,160-185) around the reference's deinterleave_samples: planar channels of
    frames + `pad` elements, zeroed first — the padding the resampler's ix + 1 read relies on."""
    a = np.ascontiguousarray(interleaved)
    frames, ch = a.shape
    outs = [np.zeros(frames + pad, dtype=a.dtype) for _ in range(ch)]
    ptrs = (C.c_void_p * ch)(*[o.ctypes.data for o in outs])
    R = ref_deint()
    if a.dtype == np.float32 and count_bits == 64:
        w = R.ref_deinterleave_f32(ptrs, a.ctypes.data, frames, ch, chunk)
    else:
        w = R.ref_deinterleave(ptrs, a.ctypes.data, frames, ch, a.dtype.itemsize, chunk, count_bits)
    assert w == frames, (w, frames)
    return outs


_ref_vu = None
REF_VU_PATH = os.path.join(ORACLE_DIR, "_ref", "libwbref_vu.so")


def ref_vu() -> Optional[C.CDLL]:
    """The reference's own VUMeter (oracle/Makefile: the struct cut out of engine/vu_meter.h where it lies and compiled
    unmodified; None where oracle/_ref was never built)."""
    global _ref_vu
    if _ref_vu is None:
        if not build_ref() or not os.path.exists(REF_VU_PATH):
            return None
        R = C.CDLL(REF_VU_PATH)
        R.ref_vu_push_blocks.argtypes = [c_f32p, C.c_uint32, C.c_uint32, C.c_uint32, c_f32p]
        _ref_vu = R
    return _ref_vu


def oracle_vu_levels(blocks: np.ndarray, reset_every: int = 0) -> np.ndarray:
    """what wb_oracle.c does per track and channel at the end of Track::process (track.cpp:732 -> vu_meter.h:20-30):
    p = wbo_abs_max(block); if (level < p) level = p — `blocks` is [n_blocks][n] fp32; the level after every block"""
    L = lib()
    L.wbo_abs_max.restype = C.c_float
    L.wbo_abs_max.argtypes = [c_f32p, C.c_uint32]
    level = np.float32(0.0)
    out = np.zeros(len(blocks), np.float32)
    for b, blk in enumerate(blocks):
        if reset_every and b and b % reset_every == 0:
            level = np.float32(0.0)
        a = np.ascontiguousarray(blk, dtype=np.float32)
        p = np.float32(L.wbo_abs_max(a.ctypes.data_as(c_f32p), len(a)))
        if level < p:
            level = p
        out[b] = level
    return out


def ref_vu_levels(blocks: np.ndarray, reset_every: int = 0) -> np.ndarray:
    a = np.ascontiguousarray(blocks, dtype=np.float32)
    out = np.zeros(len(a), np.float32)
    ref_vu().ref_vu_push_blocks(a.ctypes.data_as(c_f32p), a.shape[1], a.shape[0], reset_every, out.ctypes.data_as(c_f32p))
    return out


# ---------------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------------

def planar_ptrs(arrs: Sequence[np.ndarray], ctype=c_f32p):
    arr_t = ctype * len(arrs)
    return arr_t(*[a.ctypes.data_as(ctype) for a in arrs])


def void_ptrs(arrs: Sequence[np.ndarray]):
    arr_t = C.c_void_p * len(arrs)
    return arr_t(*[a.ctypes.data for a in arrs])


def f32_bits(x) -> int:
    return int(np.float32(x).view(np.uint32))


def f64_bits(x) -> int:
    return int(np.float64(x).view(np.uint64))


class OracleSampler:
    """wbo_sampler + one sample, for direct Sampler::stream comparisons."""

    def __init__(self, fmt: str, channels: int, rate: int, count: int, data: List[np.ndarray]):
        L = lib()
        self.fmt, self.channels, self.rate, self.count, self.data = fmt, channels, rate, count, data
        self._ptrs = void_ptrs(data)

        class _Sample(C.Structure):
            _fields_ = [("format", C.c_int), ("channels", C.c_uint32), ("sample_rate", C.c_uint32),
                        ("count", C.c_size_t), ("data", c_voidpp)]
        self._S = _Sample(FMT[fmt], channels, rate, count, C.cast(self._ptrs, c_voidpp))
        self.state = _Sampler(0.0, 0.0)
        L.wbo_sampler_stream.argtypes = [C.POINTER(_Sampler), C.POINTER(_Sample), C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_float, c_f32pp]

    def reset(self, off, speed, dst_rate):
        lib().wbo_sampler_reset(C.byref(self.state), off, speed, float(self.rate), float(dst_rate))

    def stream(self, out: List[np.ndarray], n: int, boff: int, gain: float):
        lib().wbo_sampler_stream(C.byref(self.state), C.byref(self._S), len(out), n, boff, np.float32(gain),
                                 planar_ptrs(out))


class OracleEngine:
    """Python handle on wbo_engine (the restated Engine/Track)."""

    def __init__(self, channels=2, block=512, sample_rate=48000):
        self.L = lib()
        self.e = self.L.wbo_engine_create(channels, block, sample_rate)
        self.C, self.F = channels, block
        self._keep = []

    def close(self):
        if self.e:
            self.L.wbo_engine_destroy(self.e)
            self.e = None

    def __del__(self):
        self.close()

    def set_bpm(self, bpm): self.L.wbo_engine_set_bpm(self.e, bpm)
    def set_playhead(self, beat): self.L.wbo_engine_set_playhead(self.e, beat)
    def set_buses(self, n): self.L.wbo_engine_set_buses(self.e, n)

    def add_sample(self, fmt, channels, rate, count, data: List[np.ndarray]) -> int:
        ptrs = void_ptrs(data)
        self._keep.append((data, ptrs))
        return self.L.wbo_engine_add_sample(self.e, FMT[fmt], channels, rate, count, C.cast(ptrs, c_voidpp))

    def add_track(self) -> int: return self.L.wbo_engine_add_track(self.e)
    def set_volume(self, t, db): self.L.wbo_track_set_volume(self.e, t, np.float32(db))
    def set_pan(self, t, p): self.L.wbo_track_set_pan(self.e, t, np.float32(p))
    def set_mute(self, t, m): self.L.wbo_track_set_mute(self.e, t, int(m))
    def set_bus(self, t, b): self.L.wbo_track_set_bus(self.e, t, b)

    def add_audio_clip(self, t, mn, mx, start_offset, sample, speed=1.0, gain=1.0) -> int:
        return self.L.wbo_engine_add_audio_clip(self.e, t, mn, mx, start_offset, sample, speed, np.float32(gain))

    def move_clip(self, t, clip, rel): return self.L.wbo_engine_move_clip(self.e, t, clip, rel)

    def resize_clip(self, t, clip, rel, resize_limit, min_length, left_side, shift=False, stretch=False):
        return self.L.wbo_engine_resize_clip(self.e, t, clip, rel, resize_limit, min_length, int(left_side), int(shift),
                                             int(stretch))

    def delete_clip(self, t, clip): return self.L.wbo_engine_delete_clip(self.e, t, clip)
    def set_clip_gain(self, t, clip, gain): return self.L.wbo_engine_set_clip_gain(self.e, t, clip, np.float32(gain))
    def delete_region(self, t, mn, mx): return self.L.wbo_engine_delete_region(self.e, t, mn, mx)

    def clips(self, t):
        """(min_time, max_time, start_offset, speed, gain, sample) of the track's sorted clip list"""
        n = self.L.wbo_track_clip_count(self.e, t)
        out = []
        for i in range(n):
            c = self.L.wbo_track_clip(self.e, t, i).contents
            out.append((c.min_time, c.max_time, c.start_offset, c.speed, c.gain, c.sample))
        return out

    def delete_track(self, slot): self.L.wbo_engine_delete_track(self.e, slot)
    def move_track(self, a, b): self.L.wbo_engine_move_track(self.e, a, b)
    def solo_track(self, slot): self.L.wbo_engine_solo_track(self.e, slot)

    def play(self): self.L.wbo_engine_play(self.e)
    def stop(self): self.L.wbo_engine_stop(self.e)

    def process(self, want_buses=False, clamp=True):
        out = [np.zeros(self.F, dtype=np.float32) for _ in range(self.C)]
        nb = self.e.contents.n_buses
        bus = np.zeros((nb, self.C, self.F), dtype=np.float32) if (want_buses and nb) else None
        self.L.wbo_engine_process_ex(self.e, planar_ptrs(out), bus.ctypes.data_as(c_f32p) if bus is not None else None,
                                     int(clamp))
        return np.stack(out), bus

    def process_from(self, running: np.ndarray, clamp=False) -> np.ndarray:
        """Engine::process continuing `running` ([C][F], the un-clamped sum of the tracks before this engine's) instead of
        the cleared output buffer — the checker for wbx_set_master_init / WBX_DIST_CHAIN."""
        out = [np.ascontiguousarray(running[c], dtype=np.float32).copy() for c in range(self.C)]
        self.L.wbo_engine_process_from.argtypes = [C.POINTER(_Engine), c_f32pp, C.c_int]
        self.L.wbo_engine_process_from(self.e, planar_ptrs(out), int(clamp))
        return np.stack(out)

    def enable_seglog(self, on=True): self.L.wbo_engine_enable_seglog(self.e, int(on))

    def seglog(self):
        """Sampler::stream calls of the last processed block:
        (track, dst_start, len, sample_offset_before, playback_speed, gain, sample)."""
        ec = self.e.contents
        return [(s.track, s.dst_start, s.len, s.sample_offset, s.playback_speed, s.gain, s.sample)
                for s in ec.seglog[:ec.n_seglog]]

    def gains(self) -> np.ndarray:
        """fl(volume*pan_c) per track as applied by the last processed block (track.cpp:728-731)."""
        n = self.e.contents.n_tracks
        g = np.zeros((n, 2), dtype=np.float32)
        for t in range(n):
            tr = self.track(t)
            vol = np.float32(0.0) if tr.mute else np.float32(tr.volume)
            g[t, 0] = vol * np.float32(tr.pan_coeffs[0])
            g[t, 1] = vol * np.float32(tr.pan_coeffs[1])
        return g

    def track(self, t) -> _Track:
        return self.e.contents.tracks[t]

    def sounding(self, t) -> bool:
        """current_audio_event of the track is a PlaySample (a Sampler::stream call follows in the next block)"""
        return self.track(t).current_event.type == 2

    def dangling(self, t) -> bool:
        """the track streams through a clip an edit has destroyed (quirk Q10): current_audio_event.clip names a pool
        chunk no live clip of the track occupies"""
        tr = self.track(t)
        if tr.current_event.type != 2:
            return False
        n = self.L.wbo_track_clip_count(self.e, t)
        return all(self.L.wbo_track_clip(self.e, t, i).contents.uid != tr.cur_clip_uid for i in range(n))

    def events(self, t):
        tr = self.track(t)
        return [(ev.type, ev.buffer_offset, ev.sample_offset, ev.speed, ev.time) for ev in tr.events[:tr.n_events]]

    def peaks(self) -> np.ndarray:
        n = self.e.contents.n_tracks
        return np.array([[self.track(t).block_peak[0], self.track(t).block_peak[1]] for t in range(n)],
                        dtype=np.float32)

    @property
    def playhead(self): return self.e.contents.playhead
    @property
    def sample_position(self): return self.e.contents.sample_position


def build_oracle_engine(spec) -> OracleEngine:
    """Build a wbo_engine from a whitebox_amd.synth.SessionSpec through the restated reference API."""
    eng = OracleEngine(spec.channels, spec.block, spec.sample_rate)
    eng.set_bpm(spec.bpm)
    if spec.playhead_start:
        eng.set_playhead(spec.playhead_start)
    if spec.n_buses:
        eng.set_buses(spec.n_buses)
    ids = []
    for i, s in enumerate(spec.samples):
        ids.append(eng.add_sample(s.fmt, s.channels, s.rate, s.frames, spec.sample_data(i)))
    for t in range(spec.n_tracks):
        eng.add_track()
        eng.set_volume(t, spec.volumes_db[t])
        eng.set_pan(t, spec.pans[t])
        if spec.mutes[t]:
            eng.set_mute(t, True)
        if spec.track_bus is not None:
            eng.set_bus(t, spec.track_bus[t])
    for c in spec.clips:
        sidx = c.sample if c.sample is not None else c.track
        rc = eng.add_audio_clip(c.track, c.min_beat, c.max_beat, c.start_offset, ids[sidx], c.speed, c.gain)
        assert rc == 0, f"oracle add_audio_clip failed rc={rc}"
    return eng


# ---- next rows: clip ingest + waveform mip-maps (restatements of dsp/sample.cpp:29-43, gfx/waveform_visual.cpp:9-246)
def oracle_deinterleave(interleaved: np.ndarray, chunk: int = 1024) -> List[np.ndarray]:
    """Sample::load_file's loop: the decoder hands `chunk` frames at a time (buffer_len_per_channel = 1024)."""
    a = np.ascontiguousarray(interleaved)
    frames, ch = a.shape
    out = [np.zeros(frames, dtype=a.dtype) for _ in range(ch)]
    ptrs = (C.c_void_p * ch)(*[o.ctypes.data for o in out])
    written = 0
    while written < frames:
        n = min(chunk, frames - written)
        part = np.ascontiguousarray(a[written:written + n])
        written = lib().wbo_deinterleave(ptrs, part.ctypes.data, n, written, ch, a.dtype.itemsize)
    return out


def oracle_mip_levels(count: int) -> int:
    return lib().wbo_mip_levels(count)


def oracle_mip(fmt: str, data: np.ndarray, level: int, quality: int) -> np.ndarray:
    n = lib().wbo_mip_data_count(len(data), level)
    out = np.zeros(n, dtype=np.int16 if quality else np.int8)
    d = np.ascontiguousarray(data)
    lib().wbo_mip_summarize(FMT[fmt], len(d), d.ctypes.data, level, 16 if quality else 8, out.ctypes.data)
    return out
