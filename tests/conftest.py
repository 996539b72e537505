import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# torch first: it ships its own HIP runtime, and libwbx.so must bind to that copy (same soname) when a test uses
# both in one process — loaded the other way round, torch finds "no HIP GPUs"
try:
    import torch  # noqa: F401
except Exception:   # pragma: no cover
    torch = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref built from /root/reference (this container only)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_ffi
    oracle_ffi.lib()
    return oracle_ffi


@pytest.fixture(scope="session")
def reflib(oracle):
    r = oracle.ref()
    if r is None:
        pytest.skip("oracle/_ref/libwbref.so not built (no /root/reference here)")
    return r
