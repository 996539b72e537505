"""The C-ABI shared library: loads without a GPU, exports every symbol include/wbx.h declares, refuses
to run without a gfx950 device (no CPU fallback), validates its arguments.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

import whitebox_amd as W
from whitebox_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "wbx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wbx_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = W.lib()
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(L, n), f"libwbx.so does not export {n}"
    # and the Python binding covers exactly the header
    assert sorted(_ffi.SYMBOLS) == names


def test_header_structs_match_binding_sizes():
    assert C.sizeof(_ffi.Config) == 40 and C.sizeof(_ffi.Segment) == 32 and C.sizeof(_ffi.PlanRecord) == 48


def test_version_and_status_strings():
    L = W.lib()
    assert L.wbx_version().startswith(b"wbx")
    assert L.wbx_status_string(0) == b"ok"
    assert L.wbx_status_string(-5) == b"no gfx950 device"
    assert L.wbx_status_string(-3) == b"unsupported"


def test_no_cpu_fallback_without_a_device():
    """Without a gfx950 device every create call fails loudly; nothing computes on the host."""
    L = W.lib()
    if L.wbx_device_count() > 0:
        pytest.skip("a GPU is visible here")
    h = C.c_void_p()
    cfg = _ffi.Config(0, 8, 1, 512, 2, 48000, 0, 0, None)
    assert L.wbx_create(C.byref(cfg), C.byref(h)) == -5 and not h.value
    assert L.wbx_engine_create(C.byref(cfg), C.byref(h)) == -5 and not h.value
    with pytest.raises(W.WbxError):
        W.Engine(8)
    with pytest.raises(W.WbxError):
        W.MixContext(8)


@pytest.mark.parametrize("field,value", [("channels", 3), ("channels", 0), ("block_frames", 510), ("block_frames", 0),
                                         ("max_tracks", 0), ("max_blocks", 0), ("max_blocks", 8192), ("sample_rate", 0)])
def test_config_validation(field, value):
    L = W.lib()
    cfg = _ffi.Config(0, 8, 1, 512, 2, 48000, 0, 0, None)
    setattr(cfg, field, value)
    h = C.c_void_p()
    assert L.wbx_create(C.byref(cfg), C.byref(h)) == -4      # WBX_ERR_INVALID, checked before the device


def test_null_arguments_are_rejected():
    L = W.lib()
    assert L.wbx_create(None, None) == -4
    assert L.wbx_sync(None) == -4
    assert L.wbx_engine_render(None, 1) == -4
    assert L.wbx_last_error(None) == b"null ctx"


def test_missing_library_is_an_import_error(monkeypatch, tmp_path):
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        _ffi.lib()


def test_adapter_audio_buffer_host_semantics(tmp_path):
    """wbx::AudioBuffer keeps the reference's construct / resize / resize_channel behaviour (the cases of the
    reference's test/test_audio_buffer.cpp).  Host-only: compiled with g++, no device call."""
    import subprocess
    exe = str(tmp_path / "audio_buffer_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(ROOT, "tests", "cpp", "audio_buffer_test.cpp"),
                           "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "whitebox_amd"), "-lwbx",
                           "-Wl,-rpath," + os.path.join(ROOT, "whitebox_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    assert "audio_buffer ok" in subprocess.check_output([exe]).decode()
