"""ctypes binding of include/wbx.h.  Loading fails loudly when libwbx.so has not been built."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("WBX_LIB") or os.path.join(_HERE, "libwbx.so")   # WBX_LIB: A/B another build of the library


class WbxError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: wbx_status {status}" + (f" ({detail})" if detail else ""))


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("max_tracks", C.c_uint32), ("max_blocks", C.c_uint32),
                ("block_frames", C.c_uint32), ("channels", C.c_uint32), ("sample_rate", C.c_uint32),
                ("group_size", C.c_uint32), ("max_segments", C.c_uint32), ("stream", C.c_void_p)]


class Segment(C.Structure):
    _fields_ = [("sample_offset", C.c_double), ("playback_speed", C.c_double), ("clip", C.c_uint32),
                ("buffer_offset", C.c_uint32), ("num_samples", C.c_uint32), ("gain", C.c_float)]


class ClipInfo(C.Structure):
    _fields_ = [("min_time", C.c_double), ("max_time", C.c_double), ("start_offset", C.c_double), ("speed", C.c_double),
                ("gain", C.c_float), ("sample", C.c_uint32)]


class PluginProcessInfo(C.Structure):
    _fields_ = [("sample_count", C.c_uint32), ("input_buffer_count", C.c_uint32), ("output_buffer_count", C.c_uint32),
                ("n_channels", C.c_uint32), ("input_buffer", C.POINTER(C.POINTER(C.c_float))),
                ("output_buffer", C.POINTER(C.POINTER(C.c_float))), ("input_event_list", C.c_void_p),
                ("sample_rate", C.c_double), ("tempo", C.c_double), ("project_time_in_ppq", C.c_double),
                ("project_time_in_samples", C.c_int64), ("playing", C.c_int32)]


class Plugin(C.Structure):
    _fields_ = [("userdata", C.c_void_p), ("process", C.c_void_p)]


class PlanRecord(C.Structure):
    _fields_ = [("block", C.c_uint32), ("track", C.c_uint32), ("buffer_offset", C.c_uint32),
                ("num_samples", C.c_uint32), ("num_actual", C.c_uint32), ("sample", C.c_uint32),
                ("sample_offset", C.c_double), ("playback_speed", C.c_double), ("gain", C.c_float),
                ("flags", C.c_uint32)]


FMT = {"i16": 3, "i24": 5, "i32": 7, "f32": 9}
OUT_FMT = {"i16": 3, "i24": 5, "i24_x8": 6, "i32": 7, "f32": 9}

# every symbol include/wbx.h declares: name -> (restype, argtypes)
_vp, _u32, _i32, _f, _d, _sz = C.c_void_p, C.c_uint32, C.c_int32, C.c_float, C.c_double, C.c_size_t
_pp = C.POINTER(C.c_void_p)
_fpp = C.POINTER(C.POINTER(C.c_float))
SYMBOLS = {
    "wbx_version": (C.c_char_p, []),
    "wbx_device_count": (C.c_int, []),
    "wbx_status_string": (C.c_char_p, [C.c_int]),
    "wbx_create": (C.c_int, [C.POINTER(Config), _pp]),
    "wbx_destroy": (None, [_vp]),
    "wbx_last_error": (C.c_char_p, [_vp]),
    "wbx_clip_upload": (C.c_int, [_vp, _u32, C.c_int, _u32, _u32, C.c_uint64, _pp]),
    "wbx_clip_synth": (C.c_int, [_vp, _u32, C.c_int, _u32, _u32, C.c_uint64, C.c_uint64, _u32, _f]),
    "wbx_clip_free": (C.c_int, [_vp, _u32]),
    "wbx_clip_pool_stats": (C.c_int, [_vp, C.POINTER(_u32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "wbx_clip_upload_interleaved": (C.c_int, [_vp, _u32, C.c_int, _u32, _u32, C.c_uint64, _vp]),
    "wbx_clip_ingest_device": (C.c_int, [_vp, _u32, C.c_int, _u32, _u32, C.c_uint64, _vp]),
    "wbx_clip_download": (C.c_int, [_vp, _u32, _u32, _vp]),
    "wbx_mip_levels": (_u32, [C.c_uint64]),
    "wbx_mip_data_count": (C.c_uint64, [C.c_uint64, _u32]),
    "wbx_clip_build_mipmaps": (C.c_int, [_vp, _u32, C.c_int]),
    "wbx_clip_fetch_mipmap": (C.c_int, [_vp, _u32, _u32, _vp]),
    "wbx_clip_mipmap_device": (C.c_int, [_vp, _u32, _u32, _pp, C.POINTER(C.c_uint64)]),
    "wbx_render_order": (C.c_int, [_vp, _u32, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(C.c_int)]),
    "wbx_device_info": (C.c_int, [_vp, C.c_char_p, _sz, C.c_char_p, _sz]),
    "wbx_set_routing": (C.c_int, [_vp, _u32, C.POINTER(_i32), _u32]),
    "wbx_submit": (C.c_int, [_vp, _u32, _u32, C.POINTER(Segment), C.POINTER(_u32), C.POINTER(_f)]),
    "wbx_fetch": (C.c_int, [_vp, _fpp, C.POINTER(_f), C.POINTER(_f)]),
    "wbx_fetch_interleaved": (C.c_int, [_vp, C.c_int, _vp]),
    "wbx_set_master_format": (C.c_int, [_vp, C.c_int]),
    "wbx_sync": (C.c_int, [_vp]),
    "wbx_render_status": (C.c_int, [_vp]),
    "wbx_master_ready": (C.c_int, [_vp, _vp]),
    "wbx_partial_master": (C.c_int, [_vp, _pp, C.POINTER(_sz)]),
    "wbx_finalize_master": (C.c_int, [_vp, _vp, _u32, C.c_int, _vp]),
    "wbx_finalize_master_into": (C.c_int, [_vp, _vp, _vp, _u32, C.c_int, _vp]),
    "wbx_set_clamp": (C.c_int, [_vp, C.c_int]),
    "wbx_set_master_target": (C.c_int, [_vp, _vp]),
    "wbx_set_master_init": (C.c_int, [_vp, _vp]),
    "wbx_shard_tracks": (None, [_u32, _u32, _u32, C.POINTER(_u32), C.POINTER(_u32)]),
    "wbx_dist_new_id": (C.c_int, [_vp]),
    "wbx_dist_init": (C.c_int, [_vp, _vp, _u32, _u32, C.c_int]),
    "wbx_dist_info": (C.c_int, [_vp, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(C.c_int)]),
    "wbx_dist_result_rank": (C.c_int, [_vp, C.POINTER(_u32)]),
    "wbx_dist_exchange": (C.c_int, [_vp, _vp]),
    "wbx_dist_exchange_time": (C.c_int, [_vp, C.POINTER(_d), C.POINTER(C.c_uint64)]),
    "wbx_dist_allgather": (C.c_int, [_vp, _vp, _vp, _sz]),
    "wbx_dist_sync": (C.c_int, [_vp]),
    "wbx_dist_barrier": (C.c_int, [_vp]),
    "wbx_dist_max": (C.c_int, [_vp, C.POINTER(_d)]),
    "wbx_dist_shutdown": (C.c_int, [_vp]),
    "wbx_pace": (C.c_int, [_vp, _u32]),
    "wbx_host_alloc": (C.c_int, [_sz, _pp]),
    "wbx_host_free": (C.c_int, [_vp]),
    "wbx_kernel_time": (C.c_int, [_vp, C.c_int, C.POINTER(_d), C.POINTER(C.c_uint64)]),
    "wbx_kernel_name": (C.c_char_p, [_vp]),
    "wbx_render_uniform_speed": (C.c_double, [_vp]),
    "wbx_xcd_count": (_u32, [_vp]),
    "wbx_tail_time": (C.c_int, [_vp, C.POINTER(_d)]),
    "wbx_gap_time": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "wbx_engine_create": (C.c_int, [C.POINTER(Config), _pp]),
    "wbx_engine_destroy": (None, [_vp]),
    "wbx_engine_set_audio_channel_config": (C.c_int, [_vp, _u32, _u32, _u32]),
    "wbx_engine_last_error": (C.c_char_p, [_vp]),
    "wbx_engine_ctx": (_vp, [_vp]),
    "wbx_engine_set_bpm": (C.c_int, [_vp, _d]),
    "wbx_engine_set_playhead_position": (C.c_int, [_vp, _d]),
    "wbx_engine_add_track": (C.c_int, [_vp, C.POINTER(_u32)]),
    "wbx_engine_set_buses": (C.c_int, [_vp, _u32]),
    "wbx_track_set_volume": (C.c_int, [_vp, _u32, _f]),
    "wbx_track_set_pan": (C.c_int, [_vp, _u32, _f]),
    "wbx_track_set_mute": (C.c_int, [_vp, _u32, C.c_int]),
    "wbx_track_set_bus": (C.c_int, [_vp, _u32, _i32]),
    "wbx_engine_add_plugin_to_track": (C.c_int, [_vp, _u32, _vp]),
    "wbx_engine_delete_plugin_from_track": (C.c_int, [_vp, _u32]),
    "wbx_track_get_plugin": (C.c_int, [_vp, _u32, _pp]),
    "wbx_engine_delete_track": (C.c_int, [_vp, _u32]),
    "wbx_engine_clear_all": (C.c_int, [_vp]),
    "wbx_engine_move_track": (C.c_int, [_vp, _u32, _u32]),
    "wbx_engine_solo_track": (C.c_int, [_vp, _u32]),
    "wbx_engine_add_sample": (C.c_int, [_vp, C.c_int, _u32, _u32, C.c_uint64, _pp, C.POINTER(_u32)]),
    "wbx_engine_add_sample_interleaved": (C.c_int, [_vp, C.c_int, _u32, _u32, C.c_uint64, _vp, C.POINTER(_u32)]),
    "wbx_engine_add_sample_synth": (C.c_int, [_vp, C.c_int, _u32, _u32, C.c_uint64, C.c_uint64, _u32, _f,
                                              C.POINTER(_u32)]),
    "wbx_engine_delete_sample": (C.c_int, [_vp, _u32]),
    "wbx_engine_add_audio_clip": (C.c_int, [_vp, _u32, _d, _d, _d, _u32, _d, _f]),
    "wbx_engine_move_clip": (C.c_int, [_vp, _u32, _u32, _d]),
    "wbx_engine_resize_clip": (C.c_int, [_vp, _u32, _u32, _d, _d, _d, C.c_int, C.c_int, C.c_int]),
    "wbx_engine_delete_clip": (C.c_int, [_vp, _u32, _u32]),
    "wbx_engine_delete_region": (C.c_int, [_vp, _u32, _d, _d]),
    "wbx_engine_set_clip_gain": (C.c_int, [_vp, _u32, _u32, _f]),
    "wbx_engine_clip_count": (C.c_int, [_vp, _u32, C.POINTER(_u32)]),
    "wbx_engine_get_clip": (C.c_int, [_vp, _u32, _u32, C.POINTER(ClipInfo)]),
    "wbx_calc_move_clip": (None, [_d] * 4 + [C.POINTER(_d)] * 2),
    "wbx_calc_resize_clip": (None, [_d] * 11 + [C.c_int] * 4 + [C.POINTER(_d)] * 4),
    "wbx_calc_clip_shift": (_d, [_d] * 4),
    "wbx_shift_clip_content": (_d, [_d] * 5),
    "wbx_engine_play": (C.c_int, [_vp]),
    "wbx_engine_stop": (C.c_int, [_vp]),
    "wbx_engine_process": (C.c_int, [_vp, _fpp]),
    "wbx_engine_process_interleaved": (C.c_int, [_vp, C.c_int, _vp]),
    "wbx_engine_render": (C.c_int, [_vp, _u32]),
    "wbx_engine_transport": (C.c_int, [_vp, C.POINTER(_d), C.POINTER(_d), C.POINTER(C.c_int)]),
    "wbx_engine_levels": (C.c_int, [_vp, C.POINTER(_f), _u32]),
    "wbx_engine_thread_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _u32]),
    "wbx_engine_sequencer_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "wbx_engine_callback_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "wbx_engine_perf_usage": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "wbx_calc_perf_update": (C.c_double, [C.c_double] * 3),
    "wbx_calc_perf_usage": (C.c_double, [C.c_double]),
    "wbx_calc_buffer_period_ms": (C.c_double, [_u32, _u32]),
    "wbx_engine_fetch_plan": (C.c_int, [_vp, C.POINTER(PlanRecord), _sz, C.POINTER(_sz)]),
}

_lib = None


def lib_path() -> str:
    return _LIB


def lib() -> C.CDLL:
    """The HIP library.  No fallback: a missing build is an error, not a slower path."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise ImportError(f"{_LIB} is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); whitebox_amd has no CPU implementation")
        L = C.CDLL(_LIB)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)          # AttributeError here = header and library disagree
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
