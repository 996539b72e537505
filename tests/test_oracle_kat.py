"""Known-answer tests pinning the oracle to the REAL reference engine.

The numbers below are the outputs the survey recorded from the reference's own Engine::process /
Track::process / Sampler::stream compiled and run in the survey container (SURVEY.md §8(a) A8 and
§8(c) "Seek-math KATs" / "Resampler KAT"), each asserted bit-for-bit.  Until round 5 they were the only pin the clip sequencer
had; since then tests/test_ref_engine.py holds the oracle's sequencer to the reference's own compiled code (oracle/_ref/
wbref_engine) — what stays KAT-pinned is Engine::reserve_track_region, which that build cannot hold.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O


def _pan(L, p):
    l, r = C.c_float(), C.c_float()
    L.wbo_pan_coefs(np.float32(p), 2, C.byref(l), C.byref(r))
    return O.f32_bits(l.value), O.f32_bits(r.value)


def test_pan_law_kats(oracle):
    L = oracle.lib()
    assert _pan(L, 0.0) == (0x3F800000, 0x3F800000)
    assert _pan(L, 0.3) == (0x3F3D2A29, 0x3F9A5828)
    assert _pan(L, 1.0)[1] == 0x3FB504F3
    assert _pan(L, -1.0)[0] == 0x3FB504F3
    # p=+1: left = sin(0)*sqrt2 = 0 exactly; p=-1: right = sin(0) = 0
    assert _pan(L, 1.0)[0] == 0 and _pan(L, -1.0)[1] == 0


def test_unimplemented_pan_laws_return_zero(oracle):
    """Q7: Balanced / 4.5 dB / 6 dB laws compute nothing (panning_law.cpp:16-28)."""
    L = oracle.lib()
    for law in (1, 3, 4):
        l, r = C.c_float(9), C.c_float(9)
        L.wbo_pan_coefs(np.float32(0.25), law, C.byref(l), C.byref(r))
        assert (l.value, r.value) == (0.0, 0.0)


def test_db_to_linear_kats(oracle):
    L = oracle.lib()
    assert O.f32_bits(L.wbo_db_to_linear(-3.0)) == 0x3F353BEF
    assert O.f32_bits(L.wbo_db_to_linear(-6.0)) == 0x3F004DCE
    assert L.wbo_db_to_linear(0.0) == 1.0
    assert L.wbo_db_to_linear(-72.0) == 0.0          # Q8: <= -72 dB is exactly zero
    assert L.wbo_db_to_linear(-100.0) == 0.0
    assert L.wbo_db_to_linear(-71.99) > 0.0


def _ramp(n, plus_one):
    base = np.arange(n, dtype=np.float64) + (1 if plus_one else 0)
    return [np.concatenate([(0.001 * base).astype(np.float32), np.zeros(16, np.float32)]) for _ in range(2)]


def test_resampler_kat(oracle):
    """Stereo ramp 0.001*i, 44.1 k -> 48 k, gain 0.5, count = 2000."""
    s = oracle.OracleSampler("f32", 2, 44100, 2000, _ramp(2000, False))
    s.reset(0.0, 1.0, 48000)
    assert O.f64_bits(s.state.playback_speed) == 0x3FED666666666666      # 0.91874999999999996
    offs = []
    for b in range(5):
        out = [np.zeros(512, np.float32) for _ in range(2)]
        s.stream(out, 512, 0, 0.5)
        offs.append(s.state.sample_offset)
        if b == 0:
            assert out[0][1] == np.float32(0.000459375005)
            assert out[0][511] == np.float32(0.23474063)
            assert np.array_equal(out[0], out[1])
        if b == 4:      # Q4: tail writes ceil((2000-1881.6)/0.91875) = 129 frames, offset still advances fully
            assert np.count_nonzero(out[0]) == 129 and not out[0][129:].any()
    assert O.f64_bits(offs[0]) == 0x407D666666666666      # 470.39999999999998
    assert offs[1] == 940.79999999999995
    assert O.f64_bits(offs[2]) == 0x40960CCCCCCCCCCC      # 1411.1999999999998
    assert offs[4] == 2352.0
    # finished: sample_offset_ >= count -> stream returns without touching anything (sampler.cpp:99-100)
    out = [np.zeros(512, np.float32) for _ in range(2)]
    s.stream(out, 512, 0, 0.5)
    assert s.state.sample_offset == 2352.0 and not out[0].any()


def test_seek_math_kat(oracle):
    """Ramp clip 0.001*(i+1), 120 BPM, 48 k: clip at beats [100/24000, 700/24000], start_offset 10,
    gain 0.5; second clip at beat 812/24000."""
    e = oracle.OracleEngine(2, 512, 48000)
    e.set_bpm(120.0)
    s = e.add_sample("f32", 2, 48000, 4000, _ramp(4000, True))
    t = e.add_track()
    assert e.add_audio_clip(t, 100 / 24000, 700 / 24000, 10.0, s, 1.0, 0.5) == 0
    assert e.add_audio_clip(t, 812 / 24000, 2000 / 24000, 0.0, s, 1.0, 1.0) == 0
    e.play()
    b0, _ = e.process()
    ev0 = e.events(t)
    assert [(x[0], x[1], x[2]) for x in ev0] == [(O.EV_PLAY, 100, 10)]
    assert not b0[0][:100].any()
    assert b0[0][100] == np.float32(0.0055)                     # first non-zero at frame 100 of block 0
    assert e.sample_position == 512.0
    b1, _ = e.process()
    ev1 = e.events(t)
    assert [(x[0], x[1], x[2]) for x in ev1] == [(O.EV_STOP, 188, 0), (O.EV_PLAY, 300, 0)]
    assert b1[0][187] == np.float32(0.305)                      # last frame of clip 1
    assert b1[0][188] == 0.0 and not b1[0][188:300].any()
    assert b1[0][300] == np.float32(0.001)                      # clip 2 starts at frame 300 of block 1
    assert np.array_equal(b0[0], b0[1]) and np.array_equal(b1[0], b1[1])
    assert e.sample_position == 1024.0
    e.close()


def test_c1_plumbing_mono_unity(oracle):
    """BASELINE config 1: 8 mono tracks, unity gain: out[c] = sequential sum of the 8 clips, both channels."""
    from whitebox_amd import synth
    spec = synth.make_session("c1", 8, clip_channels=1, n_blocks=4, unity_gain=True, seed=0x5EED0001)
    e = oracle.build_oracle_engine(spec)
    e.play()
    for b in range(4):
        out, _ = e.process()
        acc = np.zeros(512, np.float32)
        for t in range(8):
            acc = acc + spec.sample_data(t)[0][b * 512:(b + 1) * 512]
        assert np.array_equal(out[0], acc) and np.array_equal(out[1], acc)
        g = e.gains()
        assert (g.view(np.uint32) == 0x3F800000).all()          # gL = gR = 1.0 exactly
    e.close()


def test_not_playing_is_silent(oracle):
    from whitebox_amd import synth
    spec = synth.make_session("idle", 4, n_blocks=2)
    e = oracle.build_oracle_engine(spec)
    out, _ = e.process()
    assert not out.any() and e.playhead == 0.0 and e.sample_position == 0.0
    e.close()
