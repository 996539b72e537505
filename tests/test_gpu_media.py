"""GPU parity tests for the two rows next to the mix path, through the C ABI:
  * clip ingest: interleaved decoder frames -> planar clip storage (deinterleave_samples, dsp/sample.cpp:29-43)
  * waveform mip-maps (summarize_for_mipmaps_impl / WaveformVisual::create, gfx/waveform_visual.cpp:9-246)
Byte / integer work: everything is compared bit-for-bit with the oracle's restatement."""
import numpy as np
import pytest

import oracle_ffi as O
import whitebox_amd as W
from whitebox_amd import synth
from whitebox_amd.engine import build_engine

pytestmark = pytest.mark.gpu

DT = {"i16": np.int16, "i32": np.int32, "f32": np.float32}


def make_pcm(fmt, frames, ch, seed):
    rng = np.random.default_rng(seed)
    if fmt == "i16":
        return rng.integers(-32768, 32768, (frames, ch)).astype(np.int16)
    if fmt == "i32":
        return rng.integers(-2**31, 2**31, (frames, ch)).astype(np.int32)
    return (rng.standard_normal((frames, ch)) * 0.4).astype(np.float32)


@pytest.mark.parametrize("fmt", ["i16", "i32", "f32"])
@pytest.mark.parametrize("ch", [1, 2])
@pytest.mark.parametrize("frames", [1, 3, 1024, 2501, 70001])
def test_ingest_matches_oracle(fmt, ch, frames):
    a = make_pcm(fmt, frames, ch, frames * 3 + ch)
    exp = O.oracle_deinterleave(a)
    ctx = W.MixContext(4)
    ctx.clip_upload_interleaved(0, fmt, 48000, a)
    for c in range(ch):
        got = ctx.clip_download(0, c, frames, DT[fmt])
        assert np.array_equal(got.view(np.uint8), exp[c].view(np.uint8))
    ctx.close()


def test_ingest_multi_chunk_and_device_source():
    """longer than one 4 Mi-frame staging chunk (both staging buffers in play), and the device-source entry point"""
    torch = pytest.importorskip("torch")
    frames, ch = (4 << 20) * 2 + 12345, 2
    a = make_pcm("i16", frames, ch, 99)
    ctx = W.MixContext(4)
    ctx.clip_upload_interleaved(0, "i16", 44100, a)
    for c in range(ch):
        assert np.array_equal(ctx.clip_download(0, c, frames, np.int16), a[:, c])
    d = torch.from_numpy(a).cuda()
    torch.cuda.synchronize()
    ctx.clip_ingest_device(1, "i16", ch, 44100, frames, d.data_ptr())
    for c in range(ch):
        assert np.array_equal(ctx.clip_download(1, c, frames, np.int16), a[:, c])
    # a misaligned device pointer is refused, not mis-read
    with pytest.raises(W.WbxError):
        ctx.clip_ingest_device(2, "i16", ch, 44100, frames - 8, d.data_ptr() + 4)
    ctx.close()


def test_ingested_clip_plays_like_planar_upload():
    """the whole path: an engine fed through add_sample_interleaved renders the same bits as through add_sample"""
    spec = synth.make_session("ing", 24, seek=True, src_rate=44100, n_blocks=6, seed=0x1A)
    outs = []
    for mode in ("planar", "interleaved"):
        eng = build_engine(spec, max_blocks=6, device_synth=False, interleaved_ingest=(mode == "interleaved"))
        eng.play()
        eng.render(6)
        m, pk, _ = eng.ctx.fetch(peaks=True)
        outs.append((m.copy(), pk.copy()))
        eng.close()
    assert np.array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32))
    assert np.array_equal(outs[0][1].view(np.uint32), outs[1][1].view(np.uint32))


def special_values(fmt, data):
    if fmt == "f32":
        data[5] = 1.0
        data[9] = -1.0
        data[11] = 2.5      # beyond [-1, 1]: (T)conv wraps like the reference's x86 build
        data[12] = -3.0
        data[20] = np.nan
        data[21] = np.inf
    return data


@pytest.mark.parametrize("fmt", ["i16", "i32", "f32"])
@pytest.mark.parametrize("quality", [0, 1])
@pytest.mark.parametrize("frames", [65, 83, 2048, 2049, 8195, 100003])
def test_mipmaps_match_oracle(fmt, quality, frames):
    a = make_pcm(fmt, frames, 2, frames + quality)
    a[:, 0] = special_values(fmt, a[:, 0].copy())
    ctx = W.MixContext(4)
    ctx.clip_upload_interleaved(0, fmt, 48000, a)
    ctx.build_mipmaps(0, quality)
    levels = ctx.L.wbx_mip_levels(frames)
    assert levels == O.oracle_mip_levels(frames)
    for lvl in range(levels):
        got = ctx.fetch_mipmap(0, lvl, 2, frames, quality)
        for c in range(2):
            exp = O.oracle_mip(fmt, np.ascontiguousarray(a[:, c]), lvl, quality)
            assert got.shape[1] == len(exp)
            assert np.array_equal(got[c], exp), (fmt, quality, frames, lvl, c, np.flatnonzero(got[c] != exp)[:8])
    ctx.close()


def test_mipmaps_plateaus_and_order():
    """equal runs (first occurrence wins) and min/max order inside a pair, at every level"""
    frames = 40000
    rng = np.random.default_rng(5)
    data = np.repeat(rng.integers(-5, 6, frames // 16 + 1), 16)[:frames].astype(np.int16) * 3000
    ctx = W.MixContext(4)
    ctx.clip_upload_interleaved(0, "i16", 48000, data.reshape(-1, 1))
    for quality in (0, 1):
        ctx.build_mipmaps(0, quality)
        for lvl in range(O.oracle_mip_levels(frames)):
            got = ctx.fetch_mipmap(0, lvl, 1, frames, quality)[0]
            assert np.array_equal(got, O.oracle_mip("i16", data, lvl, quality)), (quality, lvl)
    ctx.close()


def test_mipmaps_full_size_clip():
    """a clip of the bench session's size (5.5 M frames, 9 levels): every level against the oracle, plus the
    size-independent property that a level's pair is the ordered merge of the four pairs below it"""
    frames = 5_500_003
    a = make_pcm("f32", frames, 1, 1234)
    ctx = W.MixContext(4)
    ctx.clip_upload_interleaved(0, "f32", 44100, a)
    ctx.build_mipmaps(0, 1)
    levels = ctx.L.wbx_mip_levels(frames)
    assert levels == 9
    prev = None
    for lvl in range(levels):
        got = ctx.fetch_mipmap(0, lvl, 1, frames, 1)[0]
        assert np.array_equal(got, O.oracle_mip("f32", np.ascontiguousarray(a[:, 0]), lvl, 1)), lvl
        if prev is not None:
            n = min(len(got) // 2, len(prev) // 8)       # pairs fully covered by stored pairs of the level below
            lo = prev[:n * 8].reshape(n, 8)
            assert np.array_equal(got[:2 * n].reshape(n, 2).min(axis=1), lo.min(axis=1))
            assert np.array_equal(got[:2 * n].reshape(n, 2).max(axis=1), lo.max(axis=1))
        prev = got
    ctx.close()


def test_mipmaps_reject_unsupported():
    ctx = W.MixContext(4)
    ctx.clip_upload(0, "i24", 48000, [np.zeros(100, np.int32)])
    with pytest.raises(W.WbxError):
        ctx.build_mipmaps(0, 0)       # the reference's switch has no 24-bit case
    with pytest.raises(W.WbxError):
        ctx.fetch_mipmap(0, 0, 1, 100, 0)
    ctx.close()


def test_mipmaps_match_the_reference_fixture():
    """the device's mip levels against tests/golden/mip.npz — outputs of the reference's own summarize_for_mipmaps_impl
    (oracle/gen_golden.py; every storage format, both qualities, every level, values beyond [-1, 1], ±Inf, NaN)"""
    from test_oracle_golden import mip_golden_cases
    n = 0
    for name, fmt, data, want in mip_golden_cases():
        ctx = W.MixContext(2)
        ctx.clip_upload_interleaved(0, fmt, 48000, np.ascontiguousarray(data.reshape(-1, 1)))
        for q in (0, 1):
            ctx.build_mipmaps(0, q)
            for lvl in range(ctx.L.wbx_mip_levels(len(data))):
                got = ctx.fetch_mipmap(0, lvl, 1, len(data), q)[0]
                assert np.array_equal(got, want[(q, lvl)]), (name, q, lvl, np.flatnonzero(got != want[(q, lvl)])[:8])
                n += 1
        ctx.close()
    assert n == 82


def test_ingest_matches_the_reference_fixture():
    """the device's planar clip storage against tests/golden/ingest.npz — outputs of the reference's own
    deinterleave_samples<T> under load_file's loop (oracle/gen_golden.py ingest); clips of more than two channels are refused
    (the path's AudioBuffer is mono / stereo), never mis-read"""
    from test_oracle_golden import ingest_golden_cases
    n = 0
    for name, a, want in ingest_golden_cases():
        frames, ch = a.shape
        fmt = name.split("_")[0]
        ctx = W.MixContext(2)
        if ch > 2:
            with pytest.raises(W.WbxError):
                ctx.clip_upload_interleaved(0, fmt, 48000, a)
            ctx.close()
            continue
        ctx.clip_upload_interleaved(0, fmt, 48000, a)
        for c in range(ch):
            got = ctx.clip_download(0, c, frames, DT[fmt])
            assert np.array_equal(got.view(np.uint8), want[c, :frames].view(np.uint8)), (name, c)
        ctx.close()
        n += 1
    assert n == 36
