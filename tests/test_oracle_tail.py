"""tests/oracle_tail.py against the plain oracle: sharded + sparse answers for blocks deep inside a render are, bit for
bit, what the single-threaded oracle computes with all the audio present."""
import numpy as np
import pytest

import oracle_ffi as O
import oracle_tail as OT
from whitebox_amd import synth


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def dense(spec, n_blocks):
    e = O.build_oracle_engine(spec)
    e.enable_seglog()
    e.play()
    out = {}
    for b in range(n_blocks):
        m, _ = e.process()
        rows = [(t, ds, min(ln, 0xFFFF), O.f64_bits(off), O.f64_bits(spd), O.f32_bits(g)) for (t, ds, ln, off, spd, g, smp) in e.seglog()]
        out[b] = (m, e.peaks(), rows)
    e.close()
    return out


@pytest.mark.parametrize("name,kw,cut,threads", [
    ("c3", dict(src_rate=44100), 0.0, 7), ("c4", dict(n_buses=8), 0.0, 5), ("c3cut", dict(src_rate=44100), 5.3, 4),
    ("i16r", dict(src_rate=44100, fmt="i16"), 2.7, 3), ("hot", dict(src_rate=44100, amp=0.05), 0.0, 6)])
def test_sharded_sparse_oracle_equals_the_plain_one(name, kw, cut, threads):
    N, K = 96, 40
    spec = synth.make_session(name, N, n_blocks=K, seed=0x7A11, **kw)
    if cut:
        spec = synth.cut_into_clips(spec, cut, K)
    want = dense(spec, K)
    check = [0, 1, 17, 18, 33, 39]
    got = OT.oracle_at_blocks(OT.descs_from_spec(spec), check, n_buses=spec.n_buses, threads=threads)
    for b in check:
        assert np.array_equal(bits(got[b][0]), bits(want[b][0])), (name, b)
        assert np.array_equal(bits(got[b][1]), bits(want[b][1])), (name, b)
        assert got[b][2] == want[b][2], (name, b)
