"""whitebox_amd — MI355X-native (gfx950) implementation of the whitebox per-block multitrack mix.

The product is the C-ABI shared library `libwbx.so` (include/wbx.h; sources in whitebox_amd/csrc);
this package is the thin Python binding used by tests and bench.py.  It never computes audio on the
CPU: `whitebox_amd.lib()` raises if the HIP library is missing, and engines cannot be created
without a gfx950 device.
"""
from ._ffi import WbxError, lib, lib_path  # noqa: F401
from .engine import AudioBuffer, Engine, MixContext, Track  # noqa: F401
