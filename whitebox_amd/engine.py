"""Python mirror of the reference's Engine / Track / AudioBuffer surface over the C ABI (include/wbx.h).

Same names and argument meaning as the reference (src/engine/engine.h, src/engine/track.h,
src/core/audio_buffer.h) so that the parity tests read like reference usage:

    eng = Engine(max_tracks=..., buffer_size=512, sample_rate=48000)     # set_audio_channel_config
    eng.set_bpm(120.0); t = eng.add_track("t"); t.set_volume(-3.0); t.set_pan(0.3)
    eng.add_audio_clip(t, "clip", min_time, max_time, start_offset, sample, speed=1.0, gain=1.0)
    eng.play(); eng.process(inp, out, 48000.0)

All arithmetic happens in libwbx.so on the GPU; this file only marshals.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _ffi
from ._ffi import WbxError


class AudioBuffer:
    """Planar fp32 buffer with the reference's fields (core/audio_buffer.h:14-175): n_samples, n_channels,
    channel_buffers; clear() and mix() with the reference's meaning (host-side containers only)."""

    def __init__(self, sample_count: int = 0, channel_count: int = 0):
        self.n_samples = sample_count
        self.n_channels = channel_count
        self.channel_buffers = [np.zeros(sample_count, dtype=np.float32) for _ in range(channel_count)]

    def get_write_pointer(self, channel: int, sample_offset: int = 0) -> np.ndarray:
        assert channel < self.n_channels, "Channel out of range"
        return self.channel_buffers[channel][sample_offset:]

    get_read_pointer = get_write_pointer

    def clear(self) -> None:
        for b in self.channel_buffers:
            b[:] = 0

    def resize(self, samples: int, clear: bool = False) -> None:
        if samples == self.n_samples:
            return
        new = [np.zeros(samples, dtype=np.float32) for _ in range(self.n_channels)]
        if not clear:
            n = min(samples, self.n_samples)
            for a, b in zip(new, self.channel_buffers):
                a[:n] = b[:n]
        self.channel_buffers, self.n_samples = new, samples

    def resize_channel(self, channel_count: int) -> None:
        assert self.n_samples != 0
        while len(self.channel_buffers) < channel_count:
            self.channel_buffers.append(np.zeros(self.n_samples, dtype=np.float32))
        del self.channel_buffers[channel_count:]
        self.n_channels = channel_count

    def _ptrs(self):
        # (cached: the audio callback calls process() with the same buffer every block)
        key = tuple(b.ctypes.data for b in self.channel_buffers)
        if getattr(self, "_ptr_key", None) != key:
            arr_t = C.POINTER(C.c_float) * self.n_channels
            self._ptr_arr = arr_t(*[b.ctypes.data_as(C.POINTER(C.c_float)) for b in self.channel_buffers])
            self._ptr_key = key
        return self._ptr_arr


def _check(st: int, where: str, handle=None, engine: bool = False):
    if st != 0:
        L = _ffi.lib()
        detail = ""
        if handle:
            detail = (L.wbx_engine_last_error(handle) if engine else L.wbx_last_error(handle)).decode()
        raise WbxError(st, where, detail or L.wbx_status_string(st).decode())


def _config(device, max_tracks, max_blocks, block, channels, sample_rate, group_size, max_segments, stream):
    return _ffi.Config(device, max_tracks, max_blocks, block, channels, sample_rate, group_size, max_segments, stream)


class MixContext:
    """Layer 1 (wbx_ctx): clips in HBM + host-sequenced segments -> master / peaks / buses."""

    def __init__(self, max_tracks: int, max_blocks: int = 1, block: int = 512, channels: int = 2,
                 sample_rate: int = 48000, group_size: int = 0, device: int = 0, max_segments: int = 0,
                 stream: Optional[int] = None, _handle=None):
        self.L = _ffi.lib()
        self.block, self.channels, self.sample_rate = block, channels, sample_rate
        self._owned = _handle is None
        if _handle is None:
            h = C.c_void_p()
            cfg = _config(device, max_tracks, max_blocks, block, channels, sample_rate, group_size, max_segments, stream)
            _check(self.L.wbx_create(C.byref(cfg), C.byref(h)), "wbx_create")
            self.h = h
        else:
            self.h = _handle
        self.last = (0, 0)
        self.n_buses = 0

    def close(self):
        if self._owned and self.h:
            self.L.wbx_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clip_upload(self, clip: int, fmt: str, rate: int, data: Sequence[np.ndarray], frames: Optional[int] = None):
        frames = len(data[0]) if frames is None else frames
        ptrs = (C.c_void_p * len(data))(*[a.ctypes.data for a in data])
        _check(self.L.wbx_clip_upload(self.h, clip, _ffi.FMT[fmt], len(data), rate, frames, ptrs), "wbx_clip_upload", self.h)

    def clip_upload_interleaved(self, clip: int, fmt: str, rate: int, frames_by_channels: np.ndarray):
        """Decoder output [frames][channels] -> planar clip storage (deinterleave_samples, dsp/sample.cpp:29-43)."""
        a = np.ascontiguousarray(frames_by_channels)
        _check(self.L.wbx_clip_upload_interleaved(self.h, clip, _ffi.FMT[fmt], a.shape[1], rate, a.shape[0], a.ctypes.data),
               "wbx_clip_upload_interleaved", self.h)

    def clip_ingest_device(self, clip: int, fmt: str, channels: int, rate: int, frames: int, device_ptr: int):
        _check(self.L.wbx_clip_ingest_device(self.h, clip, _ffi.FMT[fmt], channels, rate, frames, device_ptr),
               "wbx_clip_ingest_device", self.h)

    def clip_download(self, clip: int, channel: int, frames: int, dtype) -> np.ndarray:
        out = np.empty(frames, dtype=dtype)
        _check(self.L.wbx_clip_download(self.h, clip, channel, out.ctypes.data), "wbx_clip_download", self.h)
        return out

    def build_mipmaps(self, clip: int, quality: int):
        """WaveformVisual::create (gfx/waveform_visual.cpp:181-246): quality 0 = Low (int8), 1 = High (int16)."""
        _check(self.L.wbx_clip_build_mipmaps(self.h, clip, quality), "wbx_clip_build_mipmaps", self.h)

    def fetch_mipmap(self, clip: int, level: int, channels: int, frames: int, quality: int) -> np.ndarray:
        n = self.L.wbx_mip_data_count(frames, level)
        out = np.empty((channels, n), dtype=np.int16 if quality else np.int8)
        _check(self.L.wbx_clip_fetch_mipmap(self.h, clip, level, out.ctypes.data), "wbx_clip_fetch_mipmap", self.h)
        return out

    def clip_synth(self, clip: int, fmt: str, channels: int, rate: int, frames: int, seed: int, key_track: int, amp: float):
        _check(self.L.wbx_clip_synth(self.h, clip, _ffi.FMT[fmt], channels, rate, frames, seed, key_track,
                                     np.float32(amp)), "wbx_clip_synth", self.h)

    def set_routing(self, track_bus: Optional[Sequence[int]], n_buses: int, n_tracks: int):
        if track_bus is None or not n_buses:
            _check(self.L.wbx_set_routing(self.h, n_tracks, None, 0), "wbx_set_routing", self.h)
            self.n_buses = 0
        else:
            arr = (C.c_int32 * n_tracks)(*track_bus)
            _check(self.L.wbx_set_routing(self.h, n_tracks, arr, n_buses), "wbx_set_routing", self.h)
            self.n_buses = n_buses

    def set_clamp(self, on: bool):
        _check(self.L.wbx_set_clamp(self.h, int(on)), "wbx_set_clamp", self.h)

    def submit(self, n_blocks: int, n_tracks: int, segs: Sequence[tuple], seg_offsets: np.ndarray, gains: np.ndarray):
        """segs: (sample_offset, playback_speed, clip, buffer_offset, num_samples, gain)"""
        arr = (_ffi.Segment * max(1, len(segs)))()
        for i, s in enumerate(segs):
            arr[i] = _ffi.Segment(*s)
        so = np.ascontiguousarray(seg_offsets, dtype=np.uint32)
        g = np.ascontiguousarray(gains, dtype=np.float32)
        assert so.size == n_blocks * n_tracks + 1 and g.size == n_blocks * n_tracks * 2
        _check(self.L.wbx_submit(self.h, n_blocks, n_tracks, arr, so.ctypes.data_as(C.POINTER(C.c_uint32)),
                                 g.ctypes.data_as(C.POINTER(C.c_float))), "wbx_submit", self.h)
        self.last = (n_blocks, n_tracks)

    def fetch(self, peaks: bool = False, buses: bool = False):
        K, N = self.last
        Cn, F = self.channels, self.block
        master = [np.zeros(K * F, dtype=np.float32) for _ in range(Cn)]
        mp = (C.POINTER(C.c_float) * Cn)(*[m.ctypes.data_as(C.POINTER(C.c_float)) for m in master])
        pk = np.zeros((K, N, Cn), dtype=np.float32) if peaks else None
        bs = np.zeros((K, self.n_buses, Cn, F), dtype=np.float32) if (buses and self.n_buses) else None
        _check(self.L.wbx_fetch(self.h, mp, pk.ctypes.data_as(C.POINTER(C.c_float)) if pk is not None else None,
                                bs.ctypes.data_as(C.POINTER(C.c_float)) if bs is not None else None), "wbx_fetch", self.h)
        m = np.stack(master).reshape(Cn, K, F).transpose(1, 0, 2)     # [K][C][F]
        return m, pk, bs

    def fetch_interleaved(self, fmt: str) -> np.ndarray:
        K, _ = self.last
        dt = {"i16": np.int16, "i24": np.uint8, "i24_x8": np.int32, "i32": np.int32, "f32": np.float32}[fmt]
        out = np.zeros(K * self.block * self.channels * (3 if fmt == "i24" else 1), dtype=dt)
        _check(self.L.wbx_fetch_interleaved(self.h, _ffi.OUT_FMT[fmt], out.ctypes.data), "wbx_fetch_interleaved", self.h)
        return out

    def set_master_format(self, fmt: Optional[str]):
        """Later renders leave their master as interleaved samples of `fmt` ("i16", "i24", "i24_x8", "i32", "f32"; None:
        planar fp32 again) — the conversion runs as the sum kernel's epilogue."""
        _check(self.L.wbx_set_master_format(self.h, _ffi.OUT_FMT[fmt] if fmt else 0), "wbx_set_master_format", self.h)

    def sync(self):
        _check(self.L.wbx_sync(self.h), "wbx_sync", self.h)

    def render_status(self):
        """wait, then raise if a chained render since the last report lost a hand-over (its master is invalid): what a host
        that reads its own master target instead of fetching must call before it trusts the buffer"""
        _check(self.L.wbx_render_status(self.h), "wbx_render_status", self.h)

    def master_ready(self, stream: Optional[int] = None):
        """Order `stream` (None: the ctx stream) after the kernels that write the last render's results."""
        _check(self.L.wbx_master_ready(self.h, stream), "wbx_master_ready", self.h)

    def set_master_target(self, device_ptr: Optional[int]):
        _check(self.L.wbx_set_master_target(self.h, device_ptr), "wbx_set_master_target", self.h)

    def set_master_init(self, device_ptr: Optional[int]):
        """Later renders continue the running (un-clamped) sum in `device_ptr` instead of starting from zero."""
        _check(self.L.wbx_set_master_init(self.h, device_ptr), "wbx_set_master_init", self.h)

    def render_order(self, n_blocks: int):
        """(groups per block, longest group, is it the reference's order) for a render of n_blocks blocks"""
        ng, lg, ref = C.c_uint32(), C.c_uint32(), C.c_int()
        _check(self.L.wbx_render_order(self.h, n_blocks, C.byref(ng), C.byref(lg), C.byref(ref)), "wbx_render_order", self.h)
        return ng.value, lg.value, bool(ref.value)

    def device_info(self):
        pci, name = C.create_string_buffer(32), C.create_string_buffer(128)
        _check(self.L.wbx_device_info(self.h, pci, 32, name, 128), "wbx_device_info", self.h)
        return {"pci": pci.value.decode(), "name": name.value.decode()}

    def partial_master(self):
        p, n = C.c_void_p(), C.c_size_t()
        _check(self.L.wbx_partial_master(self.h, C.byref(p), C.byref(n)), "wbx_partial_master", self.h)
        return p.value, n.value

    def finalize_master(self, device_ptr: int, n_blocks: int, clamp: bool = True, stream: Optional[int] = None):
        _check(self.L.wbx_finalize_master(self.h, device_ptr, n_blocks, int(clamp), stream), "wbx_finalize_master", self.h)

    def finalize_master_into(self, device_ptr: int, dst_ptr: int, n_blocks: int, clamp: bool = True, stream: Optional[int] = None):
        """Clamp the reduced master out of place; `dst_ptr` may be pinned host memory."""
        _check(self.L.wbx_finalize_master_into(self.h, device_ptr, dst_ptr, n_blocks, int(clamp), stream),
               "wbx_finalize_master_into", self.h)

    def kernel_time(self, reset: bool = False):
        ms, n = C.c_double(), C.c_uint64()
        _check(self.L.wbx_kernel_time(self.h, int(reset), C.byref(ms), C.byref(n)), "wbx_kernel_time", self.h)
        return ms.value, n.value


    def kernel_name(self) -> str:
        """The mix_kernel instance the last render launched, as rocprofv3 prints it."""
        return (self.L.wbx_kernel_name(self.h) or b"").decode()

    def xcd_count(self) -> int:
        """XCDs the workgroup ids of a launch are dealt to round-robin (wbx_create's probe); 0: no such layout"""
        return int(self.L.wbx_xcd_count(self.h))

    def uniform_speed(self) -> float:
        """MixArgs::uniform_speed of the last render (0.0: no single resampling ratio)."""
        return float(self.L.wbx_render_uniform_speed(self.h))

    def tail_time(self) -> float:
        ms = C.c_double()
        _check(self.L.wbx_tail_time(self.h, C.byref(ms)), "wbx_tail_time", self.h)
        return ms.value

    def gap_time(self):
        """(mean idle time in ms between two consecutive mix launches since the last kernel_time(reset=True), pairs timed)"""
        ms, n = C.c_double(), C.c_uint64()
        _check(self.L.wbx_gap_time(self.h, C.byref(ms), C.byref(n)), "wbx_gap_time", self.h)
        return ms.value, int(n.value)


class Track:
    """wb::Track surface used by the mix path (track.h:137-139)."""

    def __init__(self, engine: "Engine", index: int, name: str):
        self.engine, self.index, self.name = engine, index, name

    def set_volume(self, db: float):
        _check(self.engine.L.wbx_track_set_volume(self.engine.h, self.index, np.float32(db)), "Track::set_volume", self.engine.h, True)

    def set_pan(self, pan: float):
        _check(self.engine.L.wbx_track_set_pan(self.engine.h, self.index, np.float32(pan)), "Track::set_pan", self.engine.h, True)

    def set_mute(self, mute: bool):
        _check(self.engine.L.wbx_track_set_mute(self.engine.h, self.index, int(mute)), "Track::set_mute", self.engine.h, True)

    def set_bus(self, bus: int):
        _check(self.engine.L.wbx_track_set_bus(self.engine.h, self.index, bus), "Track::set_bus", self.engine.h, True)

    @property
    def plugin_instance(self):
        """Track::plugin_instance (track.h:124): the effect slot — always empty (None) in this path."""
        p = C.c_void_p()
        _check(self.engine.L.wbx_track_get_plugin(self.engine.h, self.index, C.byref(p)), "wbx_track_get_plugin",
               self.engine.h, True)
        return p.value


class Engine:
    """wb::Engine surface (engine.h:29-273) for the mix path."""

    def __init__(self, max_tracks: int, buffer_size: int = 512, sample_rate: int = 48000, output_channels: int = 2,
                 max_blocks: int = 1, group_size: int = 0, device: int = 0, max_segments: int = 0,
                 stream: Optional[int] = None):
        self.L = _ffi.lib()
        h = C.c_void_p()
        cfg = _config(device, max_tracks, max_blocks, buffer_size, output_channels, sample_rate, group_size,
                      max_segments, stream)
        _check(self.L.wbx_engine_create(C.byref(cfg), C.byref(h)), "wbx_engine_create")
        self.h = h
        self.audio_buffer_size, self.audio_sample_rate, self.num_output_channels = buffer_size, sample_rate, output_channels
        self.tracks: List[Track] = []
        self.ctx = MixContext(max_tracks, max_blocks, buffer_size, output_channels, sample_rate,
                              _handle=C.c_void_p(self.L.wbx_engine_ctx(h)))
        self.n_buses = 0

    def close(self):
        if self.h:
            self.L.wbx_engine_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference-shaped surface ----
    def set_audio_channel_config(self, input_channels: int, output_channels: int, buffer_size: int, sample_rate: int):
        """Engine::set_audio_channel_config (engine.cpp:43-57) on the live engine: tracks and clips stay."""
        _check(self.L.wbx_engine_set_audio_channel_config(self.h, output_channels, buffer_size, sample_rate),
               "Engine::set_audio_channel_config", self.h, True)
        self.audio_buffer_size, self.audio_sample_rate, self.num_output_channels = buffer_size, sample_rate, output_channels
        self.ctx.block, self.ctx.channels, self.ctx.sample_rate = buffer_size, output_channels, sample_rate

    def set_bpm(self, bpm: float):
        _check(self.L.wbx_engine_set_bpm(self.h, bpm), "Engine::set_bpm", self.h, True)

    def set_playhead_position(self, beat: float):
        _check(self.L.wbx_engine_set_playhead_position(self.h, beat), "Engine::set_playhead_position", self.h, True)

    def add_track(self, name: str = "") -> Track:
        idx = C.c_uint32()
        _check(self.L.wbx_engine_add_track(self.h, C.byref(idx)), "Engine::add_track", self.h, True)
        t = Track(self, idx.value, name)
        self.tracks.append(t)
        return t

    def set_buses(self, n: int):
        _check(self.L.wbx_engine_set_buses(self.h, n), "wbx_engine_set_buses", self.h, True)
        self.n_buses = n
        self.ctx.n_buses = n

    # ---- track management (engine.cpp:210-262): the Track objects keep their identity, their slots shift ----
    def _reindex(self):
        for i, t in enumerate(self.tracks):
            t.index = i

    def delete_track(self, slot: int):
        _check(self.L.wbx_engine_delete_track(self.h, slot), "Engine::delete_track", self.h, True)
        del self.tracks[slot]
        self._reindex()

    def clear_all(self):
        _check(self.L.wbx_engine_clear_all(self.h), "Engine::clear_all", self.h, True)
        self.tracks = []

    def move_track(self, from_slot: int, to_slot: int):
        _check(self.L.wbx_engine_move_track(self.h, from_slot, to_slot), "Engine::move_track", self.h, True)
        t = self.tracks.pop(from_slot)
        self.tracks.insert(to_slot, t)
        self._reindex()

    def solo_track(self, slot: int):
        _check(self.L.wbx_engine_solo_track(self.h, slot), "Engine::solo_track", self.h, True)

    def add_plugin_to_track(self, track: Track, plugin=None):
        """Engine::add_plugin_to_track (engine.h:227): the slot is kept, processing through it is Unimplemented."""
        _check(self.L.wbx_engine_add_plugin_to_track(self.h, track.index, C.byref(plugin) if plugin is not None else None),
               "Engine::add_plugin_to_track", self.h, True)

    def delete_plugin_from_track(self, track: Track):
        _check(self.L.wbx_engine_delete_plugin_from_track(self.h, track.index), "Engine::delete_plugin_from_track", self.h, True)

    def sequencer_stats(self):
        """(renders planned by segments, tracks with a seam that did not hold, segments planned again, segments per track)"""
        out = (C.c_uint64 * 4)()
        _check(self.L.wbx_engine_sequencer_stats(self.h, out), "wbx_engine_sequencer_stats", self.h, True)
        return tuple(int(x) for x in out)

    def callback_stats(self):
        """(blocks run as one launch, ... with the sum spread over the workgroups, blocks mixed again after a give-up at the
        spread barrier, 1 when the engine has stopped spreading)"""
        out = (C.c_uint64 * 4)()
        _check(self.L.wbx_engine_callback_stats(self.h, out), "wbx_engine_callback_stats", self.h, True)
        return tuple(int(x) for x in out)

    def perf_usage(self):
        """(Engine::perf_measurer.get_usage() — the callback's share of its period, exponential average, clamped to [0, 1] —,
        the wall time in ms of the last process call)"""
        u, d = C.c_double(), C.c_double()
        _check(self.L.wbx_engine_perf_usage(self.h, C.byref(u), C.byref(d)), "wbx_engine_perf_usage", self.h, True)
        return u.value, d.value

    def thread_stats(self):
        """(locked edits the last process / render had seen, per-track cumulative drained parameter messages)"""
        n = len(self.tracks)
        seen = C.c_uint64()
        dr = (C.c_uint64 * max(1, n))()
        _check(self.L.wbx_engine_thread_stats(self.h, C.byref(seen), dr, n), "wbx_engine_thread_stats", self.h, True)
        return seen.value, list(dr[:n])

    def add_sample(self, fmt: str, rate: int, data: Sequence[np.ndarray], frames: Optional[int] = None) -> int:
        """Sample asset -> HBM.  `data` planar channel arrays (padding is added by the library)."""
        frames = len(data[0]) if frames is None else frames
        ptrs = (C.c_void_p * len(data))(*[a.ctypes.data for a in data])
        sid = C.c_uint32()
        _check(self.L.wbx_engine_add_sample(self.h, _ffi.FMT[fmt], len(data), rate, frames, ptrs, C.byref(sid)),
               "wbx_engine_add_sample", self.h, True)
        return sid.value

    def add_sample_interleaved(self, fmt: str, rate: int, frames_by_channels: np.ndarray) -> int:
        """The same from a decoder's interleaved [frames][channels] buffer (Sample::load_file, dsp/sample.cpp:112-197)."""
        a = np.ascontiguousarray(frames_by_channels)
        sid = C.c_uint32()
        _check(self.L.wbx_engine_add_sample_interleaved(self.h, _ffi.FMT[fmt], a.shape[1], rate, a.shape[0], a.ctypes.data,
                                                        C.byref(sid)), "wbx_engine_add_sample_interleaved", self.h, True)
        return sid.value

    def add_sample_synth(self, fmt: str, channels: int, rate: int, frames: int, seed: int, key_track: int, amp: float) -> int:
        sid = C.c_uint32()
        _check(self.L.wbx_engine_add_sample_synth(self.h, _ffi.FMT[fmt], channels, rate, frames, seed, key_track,
                                                  np.float32(amp), C.byref(sid)), "wbx_engine_add_sample_synth", self.h, True)
        return sid.value

    def delete_sample(self, sample: int):
        _check(self.L.wbx_engine_delete_sample(self.h, sample), "wbx_engine_delete_sample", self.h, True)

    def add_audio_clip(self, track: Track, name: str, min_time: float, max_time: float, start_offset: float,
                       sample: int, speed: float = 1.0, gain: float = 1.0):
        _check(self.L.wbx_engine_add_audio_clip(self.h, track.index, min_time, max_time, start_offset, sample, speed,
                                                np.float32(gain)), "Engine::add_audio_clip", self.h, True)

    # ---- clip edits (engine.cpp:346-407,463-475,1460-1464); `clip` = index in the track's sorted clip list ----
    def move_clip(self, track: Track, clip: int, relative_pos: float):
        _check(self.L.wbx_engine_move_clip(self.h, track.index, clip, relative_pos), "Engine::move_clip", self.h, True)

    def resize_clip(self, track: Track, clip: int, relative_pos: float, resize_limit: float, min_length: float,
                    left_side: bool, shift: bool = False, stretch: bool = False):
        _check(self.L.wbx_engine_resize_clip(self.h, track.index, clip, relative_pos, resize_limit, min_length,
                                             int(left_side), int(shift), int(stretch)), "Engine::resize_clip", self.h, True)

    def delete_clip(self, track: Track, clip: int):
        _check(self.L.wbx_engine_delete_clip(self.h, track.index, clip), "Engine::delete_clip", self.h, True)

    def delete_region(self, track: Track, min_time: float, max_time: float):
        _check(self.L.wbx_engine_delete_region(self.h, track.index, min_time, max_time), "Engine::delete_region", self.h, True)

    def set_clip_gain(self, track: Track, clip: int, gain: float):
        _check(self.L.wbx_engine_set_clip_gain(self.h, track.index, clip, np.float32(gain)), "Engine::set_clip_gain", self.h, True)

    def clips(self, track: Track):
        """(min_time, max_time, start_offset, speed, gain, sample) of the track's sorted clip list"""
        n = C.c_uint32()
        _check(self.L.wbx_engine_clip_count(self.h, track.index, C.byref(n)), "wbx_engine_clip_count", self.h, True)
        out = []
        for i in range(n.value):
            ci = _ffi.ClipInfo()
            _check(self.L.wbx_engine_get_clip(self.h, track.index, i, C.byref(ci)), "wbx_engine_get_clip", self.h, True)
            out.append((ci.min_time, ci.max_time, ci.start_offset, ci.speed, ci.gain, ci.sample))
        return out

    def play(self):
        _check(self.L.wbx_engine_play(self.h), "Engine::play", self.h, True)

    def stop(self):
        _check(self.L.wbx_engine_stop(self.h), "Engine::stop", self.h, True)

    def process(self, input_buffer: Optional[AudioBuffer], output_buffer: AudioBuffer, sample_rate: float):
        """Engine::process(const AudioBuffer<float>&, AudioBuffer<float>&, double) — one block."""
        assert output_buffer.n_samples == self.audio_buffer_size and output_buffer.n_channels == self.num_output_channels
        assert float(sample_rate) == float(self.audio_sample_rate)
        st = self.L.wbx_engine_process(self.h, output_buffer._ptrs())
        if st != 0:
            _check(st, "Engine::process", self.h, True)
        self.ctx.last = (1, len(self.tracks))

    def process_interleaved(self, fmt: str) -> np.ndarray:
        """Engine::process + the back end's interleave_samples_to(format) in one call: the block as F*C interleaved samples."""
        dt = {"i16": np.int16, "i24": np.uint8, "i24_x8": np.int32, "i32": np.int32, "f32": np.float32}[fmt]
        out = np.zeros(self.audio_buffer_size * self.num_output_channels * (3 if fmt == "i24" else 1), dtype=dt)
        _check(self.L.wbx_engine_process_interleaved(self.h, _ffi.OUT_FMT[fmt], out.ctypes.data), "Engine::process_interleaved",
               self.h, True)
        self.ctx.last = (1, len(self.tracks))
        return out

    def render(self, n_blocks: int):
        """K consecutive blocks in one device pass; fetch with self.ctx.fetch()."""
        _check(self.L.wbx_engine_render(self.h, n_blocks), "wbx_engine_render", self.h, True)
        self.ctx.last = (n_blocks, len(self.tracks))

    def transport(self):
        ph, sp, pl = C.c_double(), C.c_double(), C.c_int()
        _check(self.L.wbx_engine_transport(self.h, C.byref(ph), C.byref(sp), C.byref(pl)), "wbx_engine_transport", self.h, True)
        return ph.value, sp.value, bool(pl.value)

    def levels(self) -> np.ndarray:
        n = len(self.tracks)
        out = np.zeros((n, self.num_output_channels), dtype=np.float32)
        _check(self.L.wbx_engine_levels(self.h, out.ctypes.data_as(C.POINTER(C.c_float)), n), "wbx_engine_levels", self.h, True)
        return out

    def fetch_plan_array(self) -> np.ndarray:
        """the whole plan of the last render as a structured array (PLAN_DTYPE), ordered by (block, track, call): what a
        check of blocks deep inside a long render slices (millions of records: no Python list)"""
        n = C.c_size_t()
        self.L.wbx_engine_fetch_plan(self.h, None, 0, C.byref(n))
        buf = np.zeros(max(1, n.value), dtype=PLAN_DTYPE)
        _check(self.L.wbx_engine_fetch_plan(self.h, buf.ctypes.data_as(C.POINTER(_ffi.PlanRecord)), n.value, C.byref(n)),
               "wbx_engine_fetch_plan", self.h, True)
        return buf[:n.value]

    def fetch_plan(self, max_records: Optional[int] = None):
        """The Sampler::stream calls of the last render, ordered by (block, track, call); `max_records` keeps the first
        ones only (the head of a long render)."""
        n = C.c_size_t()
        if max_records is None:
            self.L.wbx_engine_fetch_plan(self.h, None, 0, C.byref(n))
        else:
            n.value = max_records
        arr = (_ffi.PlanRecord * max(1, n.value))()
        _check(self.L.wbx_engine_fetch_plan(self.h, arr, n.value, C.byref(n)), "wbx_engine_fetch_plan", self.h, True)
        got = n.value if max_records is None else min(n.value, max_records)
        return [(r.block, r.track, r.buffer_offset, r.num_samples, r.num_actual, r.sample, r.sample_offset,
                 r.playback_speed, r.gain, r.flags) for r in arr[:got]]


PLAN_DTYPE = np.dtype([("block", "<u4"), ("track", "<u4"), ("buffer_offset", "<u4"), ("num_samples", "<u4"), ("num_actual", "<u4"),
                       ("sample", "<u4"), ("sample_offset", "<f8"), ("playback_speed", "<f8"), ("gain", "<f4"), ("flags", "<u4")])


def plan_rows_of_blocks(plan: np.ndarray, blocks) -> dict:
    """{block: [(track, buffer_offset, num_samples, offset bits, speed bits, gain bits), ...]} of a fetch_plan_array result"""
    out = {}
    for b in blocks:
        lo, hi = np.searchsorted(plan["block"], [b, b + 1])
        p = plan[lo:hi]
        out[int(b)] = list(zip(p["track"].tolist(), p["buffer_offset"].tolist(), p["num_samples"].tolist(),
                               p["sample_offset"].view(np.uint64).tolist(), p["playback_speed"].view(np.uint64).tolist(),
                               p["gain"].view(np.uint32).tolist()))
    return out


def build_engine(spec, max_blocks: int = 8, group_size: int = 0, device: int = 0, device_synth: bool = False,
                 interleaved_ingest: bool = False, spare_tracks: int = 0) -> Engine:
    """Build a product Engine from a synth.SessionSpec through the reference-shaped API."""
    eng = Engine(max(spec.n_tracks, 1) + spare_tracks, spec.block, spec.sample_rate, spec.channels, max_blocks=max_blocks,
                 group_size=group_size, device=device)
    eng.set_bpm(spec.bpm)
    if spec.playhead_start:
        eng.set_playhead_position(spec.playhead_start)
    if spec.n_buses:
        eng.set_buses(spec.n_buses)
    ids = []
    for i, s in enumerate(spec.samples):
        if device_synth:
            ids.append(eng.add_sample_synth(s.fmt, s.channels, s.rate, s.frames, spec.seed, s.seed_track, s.amp))
        else:
            data = [np.ascontiguousarray(a[:s.frames]) for a in spec.sample_data(i)]
            if interleaved_ingest:   # as a decoder would deliver the file: [frames][channels]
                ids.append(eng.add_sample_interleaved(s.fmt, s.rate, np.stack(data, axis=1)))
            else:
                ids.append(eng.add_sample(s.fmt, s.rate, data, s.frames))
    for t in range(spec.n_tracks):
        tr = eng.add_track(f"t{t}")
        tr.set_volume(spec.volumes_db[t])
        tr.set_pan(spec.pans[t])
        if spec.mutes[t]:
            tr.set_mute(True)
        if spec.track_bus is not None:
            tr.set_bus(spec.track_bus[t])
    for c in spec.clips:
        sidx = c.sample if c.sample is not None else c.track
        eng.add_audio_clip(eng.tracks[c.track], "clip", c.min_beat, c.max_beat, c.start_offset, ids[sidx], c.speed, c.gain)
    return eng
