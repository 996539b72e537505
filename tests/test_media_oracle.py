"""CPU tests of the oracle's restatements for the two "next" rows: clip ingest
(dsp/sample.cpp:29-43) and waveform mip-maps (gfx/waveform_visual.cpp:9-246).  The reference holds no test or
vector for either and its TUs need libsndfile / the renderer, so these check the restatement against an
independent numpy formulation of the same published loops (parity with the reference itself: clip ingest unpinned; the
mip-maps are pinned elsewhere — tests/test_oracle_vs_ref.py::test_mip_summarize_bit_exact against the reference's own summariser,
tests/test_oracle_golden.py::test_mip_golden against its recorded outputs)."""
import numpy as np
import pytest

import oracle_ffi as O


def np_mip(fmt, data, level, quality):
    """summarize_for_mipmaps_impl written from the reference text with numpy scalars (slow, small inputs only)."""
    T = np.int16 if quality else np.int8
    tmin, tmax = np.iinfo(T).min, np.iinfo(T).max
    mip = 1 + 2 * level
    chunk, block = 1 << mip, 1 << (mip - 1)
    count = len(data)
    n = count // block
    n += n % 2
    out = np.zeros(n, dtype=T)
    for i in range(0, n, 2):
        idx = i * block
        length = min(chunk, count - idx)
        mn, mx, mni, mxi = tmax, tmin, 0, 0
        for j in range(length):
            s = data[idx + j]
            if fmt == "i16":
                k = np.float32(tmax) / np.float32(32767) if s >= 0 else np.float32(tmin) / np.float32(-32768)
                conv = np.float32(s) * k
            elif fmt == "i32":
                k = np.float64(tmax) / np.float64(2147483647) if s >= 0 else np.float64(tmin) / np.float64(-2147483648)
                conv = np.float64(s) * k
            else:
                conv = np.float32(s) * (np.float32(tmax) if s >= 0 else np.float32(-tmin))
            if not np.isfinite(conv) or conv >= 2147483648.0 or conv <= -2147483649.0:
                iv = -2147483648
            else:
                iv = int(np.trunc(conv))
            v = int(np.array([iv], dtype=np.int64).astype(T)[0])   # low bits, like the x86 register truncation
            if v < mn:
                mn, mni = v, j
            if v > mx:
                mx, mxi = v, j
        out[i], out[i + 1] = (mx, mn) if mxi < mni else (mn, mx)
    return out


@pytest.mark.parametrize("dtype,ch", [(np.int16, 1), (np.int16, 2), (np.int32, 2), (np.float32, 2), (np.float32, 1)])
def test_deinterleave_matches_transpose(dtype, ch):
    rng = np.random.default_rng(7)
    frames = 2500   # not a multiple of the decoder's 1024-frame buffer
    a = (rng.standard_normal((frames, ch)) * 1000).astype(dtype)
    out = O.oracle_deinterleave(a)
    for c in range(ch):
        assert np.array_equal(out[c], a[:, c])


def test_mip_level_counts():
    # WaveformVisual::create: levels while count / 4^l > 64; mip_data_count = count / 2^(2l) rounded up to even
    assert O.oracle_mip_levels(64) == 0
    assert O.oracle_mip_levels(65) == 1
    assert O.oracle_mip_levels(256) == 1
    assert O.oracle_mip_levels(260) == 2
    assert O.oracle_mip_levels(48000 * 10) == 7
    L = O.lib()
    assert L.wbo_mip_data_count(13, 0) == 14 and L.wbo_mip_data_count(13, 1) == 4 and L.wbo_mip_data_count(17, 1) == 4
    assert L.wbo_mip_data_count(83, 1) == 20 and L.wbo_mip_data_count(83, 2) == 6


@pytest.mark.parametrize("fmt", ["i16", "i32", "f32"])
@pytest.mark.parametrize("quality", [0, 1])
@pytest.mark.parametrize("count", [65, 83, 260, 1037])
def test_mip_restatement_vs_numpy(fmt, quality, count):
    rng = np.random.default_rng(count * 7 + quality)
    if fmt == "i16":
        data = rng.integers(-32768, 32768, count).astype(np.int16)
    elif fmt == "i32":
        data = rng.integers(-2**31, 2**31, count).astype(np.int32)
    else:
        data = (rng.standard_normal(count) * 0.5).astype(np.float32)
        data[5] = 1.0
        data[9] = -1.0
        data[11] = 2.5       # out of [-1, 1]: the (T)conv wrap of the reference's x86 build
        data[12] = -3.0
    for level in range(O.oracle_mip_levels(count)):
        got = O.oracle_mip(fmt, data, level, quality)
        exp = np_mip(fmt, data, level, quality)
        assert np.array_equal(got, exp), (fmt, quality, count, level)


def test_mip_first_occurrence_order_and_plateaus():
    # equal values: the earliest index wins for both extremes; order (first, second) follows occurrence
    data = np.array([3, 3, -5, -5, 9, 9, 9, -5] + [0] * 64, dtype=np.int16)
    lvl1 = O.oracle_mip("i16", data, 1, 1)      # chunks of 8
    assert tuple(lvl1[:2]) == (-5, 9)           # min first (index 2) then max (index 4)
    data2 = np.array([9, 3, -5, 9, -5, 0, 0, 0] + [0] * 64, dtype=np.int16)
    assert tuple(O.oracle_mip("i16", data2, 1, 1)[:2]) == (9, -5)
    flat = np.full(72, 7, dtype=np.int16)
    assert tuple(O.oracle_mip("i16", flat, 0, 1)[:2]) == (7, 7)
