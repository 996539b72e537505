"""Differential tests: the C restatement (oracle/wb_oracle.c) against the reference's OWN translation
units compiled into oracle/_ref/libwbref.so (sampler.cpp, panning_law.cpp, audio_format_conv.cpp +
header-only audio_buffer.h / dsp_ops.h / core_math.h).  Bit-for-bit.  Skipped where /root/reference
(and therefore oracle/_ref) does not exist — the committed golden vectors cover that case."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O
from whitebox_amd import synth

pytestmark = pytest.mark.ref


def test_scalars_bit_exact(oracle, reflib):
    L, R = oracle.lib(), reflib
    rng = np.random.default_rng(1)
    for p in np.concatenate([np.linspace(-1, 1, 401), rng.uniform(-1, 1, 2000)]).astype(np.float32):
        for law in range(5):
            a, b, c, d = C.c_float(), C.c_float(), C.c_float(), C.c_float()
            L.wbo_pan_coefs(p, law, C.byref(a), C.byref(b))
            R.ref_pan_coefs(p, law, C.byref(c), C.byref(d))
            assert (O.f32_bits(a.value), O.f32_bits(b.value)) == (O.f32_bits(c.value), O.f32_bits(d.value))
    for db in np.concatenate([np.linspace(-80, 12, 921), rng.uniform(-90, 24, 3000)]).astype(np.float32):
        assert O.f32_bits(L.wbo_db_to_linear(db)) == O.f32_bits(R.ref_db_to_linear(db))
    for _ in range(2000):
        beat, sr, bd = rng.uniform(0, 1000), rng.choice([44100.0, 48000.0, 96000.0]), 60.0 / rng.uniform(40, 300)
        assert L.wbo_beat_to_samples(beat, sr, bd) == R.ref_beat_to_samples(beat, sr, bd)
        assert L.wbo_samples_to_beat(beat * 1000, sr, bd) == R.ref_samples_to_beat(beat * 1000, sr, bd)


@pytest.mark.parametrize("fmt", ["f32", "i16", "i24", "i32"])
@pytest.mark.parametrize("src_rate,speed", [(48000, 1.0), (44100, 1.0), (48000, 0.5), (48000, 1.75),
                                             (96000, 1.0), (44100, 1.3333333333333333), (48000, 0.999999)])
def test_sampler_stream_bit_exact(oracle, reflib, fmt, src_rate, speed):
    """Unity and linear paths, all four storage formats, ragged segment lengths, non-zero buffer offsets,
    clip tail (Q2/Q4) and the finished state."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(repr((fmt, src_rate, speed)).encode()))
    count = 3000
    spec_s = synth.SampleSpec(seed_track=7, channels=2, rate=src_rate, frames=count, fmt=fmt, amp=0.7)
    sess = synth.SessionSpec("s", 1, 0xABCD, [spec_s], [], [0], [0], [False])
    data = sess.sample_data(0)
    s = oracle.OracleSampler(fmt, 2, src_rate, count, data)
    start = float(rng.integers(0, 40))
    s.reset(start, speed, 48000)
    ps, so = C.c_double(), C.c_double()
    reflib.ref_sampler_reset(C.byref(ps), C.byref(so), start, speed, float(src_rate), 48000.0)
    assert (ps.value, so.value) == (s.state.playback_speed, s.state.sample_offset)
    ptrs = O.void_ptrs(data)
    for it in range(14):
        n = int(rng.choice([512, 512, 1, 0, 37, 255, 300]))
        boff = int(rng.integers(0, 512 - n + 1)) if n < 512 else 0
        gain = np.float32(rng.choice([1.0, 0.5, 0.3333]))
        a = [np.zeros(512, np.float32) for _ in range(2)]
        b = [np.zeros(512, np.float32) for _ in range(2)]
        s.stream(a, n, boff, gain)
        reflib.ref_sampler_stream(C.byref(ps), C.byref(so), O.FMT[fmt], 2, src_rate, count, C.cast(ptrs, O.c_voidpp), 2,
                                  n, boff, gain, O.planar_ptrs(b))
        assert O.f64_bits(so.value) == O.f64_bits(s.state.sample_offset), (it, n)
        for c in range(2):
            assert np.array_equal(a[c].view(np.uint32), b[c].view(np.uint32)), (it, c, n, boff)


def test_mono_into_stereo_unity(oracle, reflib):
    """Unity path wraps the source channel (i % channels, sampler.cpp:111..147)."""
    data = [np.concatenate([synth.clip_channel(1, 0, 0, 1000, 0.5), np.zeros(16, np.float32)])]
    s = oracle.OracleSampler("f32", 1, 48000, 1000, data)
    s.reset(3.0, 1.0, 48000)
    ps, so = C.c_double(1.0), C.c_double(3.0)
    a = [np.zeros(512, np.float32) for _ in range(2)]
    b = [np.zeros(512, np.float32) for _ in range(2)]
    s.stream(a, 512, 0, 1.0)
    ptrs = O.void_ptrs(data)
    reflib.ref_sampler_stream(C.byref(ps), C.byref(so), O.FMT["f32"], 1, 48000, 1000, C.cast(ptrs, O.c_voidpp), 2, 512, 0,
                              np.float32(1.0), O.planar_ptrs(b))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[0], a[1])


def test_gain_absmax_bit_exact(oracle, reflib):
    rng = np.random.default_rng(3)
    L = oracle.lib()
    for _ in range(50):
        x = rng.normal(0, 0.3, 512).astype(np.float32)
        g = np.float32(rng.uniform(0, 2))
        a, b = x.copy(), x.copy()
        L.wbo_apply_gain(a.ctypes.data_as(O.c_f32p), 512, g)
        reflib.ref_apply_gain(b.ctypes.data_as(O.c_f32p), 512, g)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert L.wbo_abs_max(a.ctypes.data_as(O.c_f32p), 512) == reflib.ref_find_abs_maximum(b.ctypes.data_as(O.c_f32p), 512)


@pytest.mark.parametrize("name,kw", [
    ("c1", dict(n_tracks=8, clip_channels=1, unity_gain=True)),
    ("c2", dict(n_tracks=48)),
    ("c3", dict(n_tracks=40, src_rate=44100)),
    ("c4", dict(n_tracks=64, n_buses=8)),
    ("seek", dict(n_tracks=24, seek=True)),
    ("seek441", dict(n_tracks=24, seek=True, src_rate=44100)),
    ("hot", dict(n_tracks=16, amp=0.5)),          # clips: exercises the master clamp
])
def test_engine_blocks_match_reference_dsp(oracle, reflib, name, kw):
    """Whole blocks: oracle Engine::process == reference DSP (ref_mix_block) fed with the oracle's
    sequencing.  Master, bus sums, per-track peaks and sampler offsets, bit-for-bit."""
    from refmix import RefMixer
    kw = dict(kw)
    spec = synth.make_session(name, kw.pop("n_tracks"), n_blocks=6, **kw)
    e = oracle.build_oracle_engine(spec)
    e.enable_seglog()
    rm = RefMixer(spec)
    e.play()
    for b in range(6):
        out, bus = e.process(want_buses=True)
        ref_out, ref_bus, ref_peaks, _ = rm.block(e.seglog(), e.gains())
        assert np.array_equal(out.view(np.uint32), ref_out.view(np.uint32)), (name, b)
        assert np.array_equal(e.peaks().view(np.uint32), ref_peaks.view(np.uint32)), (name, b)
        if spec.n_buses:
            assert np.array_equal(bus.view(np.uint32), ref_bus.view(np.uint32))
    e.close()


@pytest.mark.parametrize("conv,dt,width", [("i16", np.int16, 1), ("i24_x8", np.int32, 1), ("i32", np.int32, 1),
                                           ("f32", np.float32, 1), ("i24", np.uint8, 3)])
def test_format_conversion_bit_exact(oracle, reflib, conv, dt, width):
    rng = np.random.default_rng(5)
    L = oracle.lib()
    src = [np.clip(rng.normal(0, 0.5, 600), -1, 1).astype(np.float32) for _ in range(2)]
    src[0][:6] = [1.0, -1.0, 0.0, -0.0, 0.9999999, -0.9999999]
    n, off = 512, 40
    a = np.zeros(n * 2 * width, dt)
    b = np.zeros(n * 2 * width, dt)
    getattr(L, "wbo_f32_to_interleaved_" + conv)(a.ctypes.data, O.planar_ptrs(src), off, n, 2)
    getattr(reflib, "ref_f32_to_" + conv)(b.ctypes.data, O.planar_ptrs(src), off, n, 2)
    assert np.array_equal(a.view(np.uint8), b.view(np.uint8))


def _clip_edit_cases(n=3000, seed=11):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        mn = float(rng.uniform(0, 64))
        ln = float(rng.uniform(0.01, 16))
        yield dict(mn=mn, mx=mn + ln, so=float(rng.uniform(0, 50000)), sp=float(rng.choice([1.0, 0.5, 1.25, 0.91875])),
                   sr=float(rng.choice([44100, 48000, 96000])), cnt=float(rng.integers(1000, 2000000)),
                   rel=float(rng.uniform(-8, 8)), lim=float(rng.uniform(0, 1)), minlen=float(rng.uniform(0.001, 0.5)),
                   mrp=float(rng.uniform(0, 4)), bd=60.0 / float(rng.uniform(60, 200)),
                   is_min=int(rng.integers(0, 2)), shift=int(rng.integers(0, 2)), stretch=int(rng.integers(0, 2)),
                   clampp=int(rng.integers(0, 2)))


def test_clip_edit_arithmetic_bit_exact(oracle, reflib):
    """calc_move_clip / calc_resize_clip / calc_clip_shift / shift_clip_content against the reference's
    header-only engine/clip_edit.h, bit patterns of every fp64 result."""
    L = oracle.lib()
    d = [C.c_double() for _ in range(8)]
    for k in _clip_edit_cases():
        L.wbo_calc_move_clip(k["mn"], k["mx"], k["rel"], k["mrp"], C.byref(d[0]), C.byref(d[1]))
        reflib.ref_calc_move_clip(k["mn"], k["mx"], k["rel"], k["mrp"], C.byref(d[2]), C.byref(d[3]))
        assert (O.f64_bits(d[0].value), O.f64_bits(d[1].value)) == (O.f64_bits(d[2].value), O.f64_bits(d[3].value))
        args = (k["mn"], k["mx"], k["so"], k["sp"], k["sr"], k["cnt"], k["rel"], k["lim"], k["minlen"], k["mrp"], k["bd"],
                k["is_min"], k["shift"], k["stretch"], k["clampp"])
        L.wbo_calc_resize_clip(*args, *[C.byref(x) for x in d[:4]])
        reflib.ref_calc_resize_clip(*args, *[C.byref(x) for x in d[4:]])
        assert [O.f64_bits(x.value) for x in d[:4]] == [O.f64_bits(x.value) for x in d[4:]], k
        a = L.wbo_calc_clip_shift(k["so"], k["rel"], k["bd"], k["sr"])
        b = reflib.ref_calc_clip_shift(k["so"], k["rel"], k["bd"], k["sr"])
        assert O.f64_bits(a) == O.f64_bits(b)
        a = L.wbo_shift_clip_content(k["so"], k["sp"], k["sr"], k["rel"], k["bd"])
        b = reflib.ref_shift_clip_content(k["so"], k["sp"], k["sr"], k["rel"], k["bd"])
        assert O.f64_bits(a) == O.f64_bits(b)


def test_clip_lower_bound_matches_reference_template(oracle, reflib):
    """wb::find_lower_bound (core/algorithm.h:24-40) instantiated with the clip sequencer's predicate
    (track.cpp:126-127, :206), against the oracle's restatement — including its quirk of never returning `end`."""
    import ctypes as C
    rng = np.random.default_rng(11)
    L = oracle.lib()
    for n in list(range(1, 12)) + [31, 64, 257]:
        for _ in range(20):
            times = np.sort(rng.choice(np.arange(0, 40) * 0.25, size=n, replace=True)).astype(np.float64)   # duplicates on purpose
            for v in list(rng.choice(times, size=min(n, 4))) + [-1.0, 100.0, float(times[0]), float(times[-1]),
                                                               float(times[n // 2]) + 0.125]:
                p = times.ctypes.data_as(C.POINTER(C.c_double))
                assert L.wbo_lower_bound_max_time(p, n, float(v)) == reflib.ref_find_lower_bound_max_time(p, n, float(v)), (n, v)


@pytest.mark.parametrize("seed", range(12))
def test_mip_summarize_bit_exact(oracle, seed):
    """wbo_mip_summarize against the reference's own summarize_for_mipmaps_impl (oracle/_ref/libwbref_mip.so: the function's
    text cut out of gfx/waveform_visual.cpp where it lies and compiled unmodified, oracle/Makefile) — random counts, every
    storage format, both output widths, every level; float inputs beyond [-1, 1], ±Inf, NaN"""
    if oracle.ref_mip() is None:
        pytest.skip("oracle/_ref/libwbref_mip.so not built (no /root/reference here)")
    rng = np.random.default_rng(9000 + seed)
    for _ in range(25):
        fmt = str(rng.choice(["f32", "i16", "i32"]))
        n = int(rng.choice([65, 66, 127, 128, 129, 255, 257, 1000, 4097, 30000, int(rng.integers(65, 120000))]))
        if fmt == "f32":
            d = (rng.standard_normal(n) * float(rng.choice([0.2, 0.9, 1.5, 4.0, 1e6]))).astype(np.float32)
            if rng.random() < 0.3:
                d[rng.integers(0, n, 3)] = [np.inf, -np.inf, np.nan]
            if rng.random() < 0.5:
                d[rng.integers(0, n, 5)] = [1.0, -1.0, 0.999999, -0.0, 0.0]
        elif fmt == "i16":
            d = rng.integers(-32768, 32768, n).astype(np.int16)
            d[rng.integers(0, n, 4)] = [32767, -32768, 0, -1]
        else:
            d = rng.integers(-2**31, 2**31, n).astype(np.int32)
            d[rng.integers(0, n, 4)] = [2**31 - 1, -2**31, 0, -1]
        for q in (0, 1):
            for lv in range(oracle.oracle_mip_levels(n)):
                got, exp = oracle.oracle_mip(fmt, d, lv, q), oracle.ref_mip_level(fmt, d, lv, q)
                assert np.array_equal(got, exp), (fmt, n, q, lv, np.flatnonzero(got != exp)[:8])


@pytest.mark.parametrize("seed", range(8))
def test_vu_meter_bit_exact(oracle, seed):
    """wbo_abs_max + the level update of wb_oracle.c against the reference's own VUMeter::push_samples / level (oracle/_ref/
    libwbref_vu.so: the struct cut out of engine/vu_meter.h where it lies and compiled unmodified) — NaN, ±Inf, -0.0 blocks,
    resets"""
    if oracle.ref_vu() is None:
        pytest.skip("oracle/_ref/libwbref_vu.so not built (no /root/reference here)")
    rng = np.random.default_rng(7100 + seed)
    for _ in range(60):
        n, nb = int(rng.choice([1, 7, 64, 128, 512])), int(rng.integers(1, 12))
        a = (rng.standard_normal((nb, n)) * float(rng.choice([0.0, 1e-3, 0.5, 3.0]))).astype(np.float32)
        k = int(rng.integers(0, 5))
        if k == 0:
            a[rng.integers(0, nb), rng.integers(0, n)] = np.nan
        elif k == 1:
            a[rng.integers(0, nb), rng.integers(0, n)] = np.inf
            a[rng.integers(0, nb), rng.integers(0, n)] = -np.inf
        elif k == 2:
            a[rng.integers(0, nb)] = -0.0
        elif k == 3:
            a[rng.integers(0, nb), :] = np.nan
        reset = int(rng.choice([0, 0, 2, 3]))
        got, exp = oracle.oracle_vu_levels(a, reset), oracle.ref_vu_levels(a, reset)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (k, n, reset, got, exp)


@pytest.mark.parametrize("seed", range(10))
def test_clip_chunks_follow_the_reference_pool(oracle, reflib, seed):
    """Q10's allocator statements (wb_oracle.c alloc_clip_uid / free_clip_uid: a clip's uid names its Pool<Clip> chunk) against
    the reference's own Pool<Clip> (core/memory.h:41-110, compiled into _ref): random adds, single deletes and region deletes
    that destroy several clips at once (in list order, track.cpp:170-172) on one oracle track, mirrored as allocate / free on the
    pool — every new clip's uid is the identity of the chunk the pool hands out, and a freed chunk reads zero behind its
    free-list link (the gain a dangling Clip* reads)"""
    reflib.ref_pool_script.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
    rng = np.random.default_rng(4400 + seed)
    e = oracle.OracleEngine(2, 64, 48000)
    e.set_bpm(120.0)
    sid = e.add_sample("f32", 1, 48000, 64, [np.zeros(80, np.float32)])
    e.add_track()
    ops, want = [], []          # the mirrored pool script and, per allocation, the oracle's uid
    handle_of = {}              # clip slot (its min_time / 10) -> pool handle
    n_alloc, next_slot = 0, 0

    def uids():
        n = oracle.lib().wbo_track_clip_count(e.e, 0)
        cl = [oracle.lib().wbo_track_clip(e.e, 0, i).contents for i in range(n)]
        return {int(round(c.min_time / 10.0)): c.uid for c in cl}

    for _ in range(400):
        live = sorted(handle_of)
        r = rng.random()
        if r < 0.5 or not live:
            slot = next_slot
            next_slot += 1
            assert e.add_audio_clip(0, 10.0 * slot, 10.0 * slot + 1.0, 0.0, sid, 1.0, 1.0) == 0
            n_alloc += 1
            handle_of[slot] = n_alloc
            ops.append(1)
            want.append(uids()[slot])
        elif r < 0.8:
            slot = int(rng.choice(live))
            assert e.delete_clip(0, live.index(slot)) == 0
            ops.append(-handle_of.pop(slot))
            want.append(1)
        else:
            a = int(rng.integers(0, len(live)))
            b = min(len(live) - 1, a + int(rng.integers(0, 4)))
            e.delete_region(0, 10.0 * live[a] - 0.5, 10.0 * live[b] + 1.5)
            for slot in live[a:b + 1]:          # destroyed in list order
                ops.append(-handle_of.pop(slot))
                want.append(1)
        assert sorted(uids()) == sorted(handle_of)
    out = (C.c_int * len(ops))()
    assert reflib.ref_pool_script((C.c_int * len(ops))(*ops), len(ops), out) == 0
    first = min(w for o, w in zip(ops, want) if o > 0)          # the oracle's uids count from the engine's first clip
    got = [x for x in out]
    exp = [w - first + 1 if o > 0 else 1 for o, w in zip(ops, want)]
    assert got == exp
    e.close()


@pytest.mark.parametrize("seed", range(8))
def test_deinterleave_bit_exact(oracle, seed):
    """wbo_deinterleave under load_file's loop against the reference's own deinterleave_samples<T> (oracle/_ref/
    libwbref_deint.so: the function's text cut out of dsp/sample.cpp where it lies and compiled unmodified as a member of a class
    template whose parameter is named sf_count_t, oracle/Makefile) — every storage format, 1-8 channels, lengths around the
    decoder's chunk, other chunk sizes, both count types; the 16 padding frames stay zero"""
    if oracle.ref_deint() is None:
        pytest.skip("oracle/_ref/libwbref_deint.so not built (no /root/reference here)")
    rng = np.random.default_rng(5200 + seed)
    for _ in range(40):
        ch = int(rng.integers(1, 9))
        frames = int(rng.choice([1, 2, 1023, 1024, 1025, 2047, 4096, int(rng.integers(1, 20000))]))
        chunk = int(rng.choice([1024, 1024, 1, 7, 4096]))
        kind = str(rng.choice(["i16", "i32", "f32"]))
        if kind == "i16":
            a = rng.integers(-32768, 32768, (frames, ch)).astype(np.int16)
        elif kind == "i32":
            a = rng.integers(-2**31, 2**31, (frames, ch)).astype(np.int32)
        else:
            a = rng.integers(0, 2**32, (frames, ch), dtype=np.uint64).astype(np.uint32).view(np.float32)   # any bit pattern
        got = oracle.oracle_deinterleave(a, chunk)
        for bits in (64, 32):
            exp = oracle.ref_deinterleave(a, chunk, bits)
            for c in range(ch):
                assert np.array_equal(got[c].view(np.uint8), exp[c][:frames].view(np.uint8)), (kind, ch, frames, chunk, bits, c)
                assert not exp[c][frames:].view(np.uint8).any()


def test_perf_measurer_and_block_period_bit_exact(oracle, reflib):
    """The load figure Engine::process ends with (engine.cpp:1577,1653): the oracle's statements against the reference's own
    PerformanceMeasurer::update / get_usage (core/timing.h:54-67) and period_to_ms(buffer_size_to_period()) (engine/audio_io.h:
    187-195, what engine.cpp:52 stores in audio_buffer_duration_ms) — both header-only, compiled into oracle/_ref/libwbref.so —
    on random loads (fp64 bit patterns), non-finite and out-of-range figures, a measurer followed over 2000 blocks, and every
    block size from 4 to 4096 frames at ten device rates"""
    L = oracle.lib()
    rng = np.random.default_rng(0xBEEF)
    for _ in range(4000):
        u = float(rng.choice([rng.random() * 1.5 - 0.2, 0.0, 1.0, np.inf, -np.inf, np.nan, 5e-324], p=[0.88, 0.02, 0.02, 0.02, 0.02, 0.02, 0.02]))
        d, t = float(10.0 ** (rng.random() * 7 - 4)), float(rng.choice([1.3, 2.9, 5.3, 10.0, 10.666666666666666, 21.3]))
        assert oracle.f64_bits(L.wbo_perf_update(u, d, t)) == oracle.f64_bits(reflib.ref_perf_update(u, d, t)), (u, d, t)
        assert oracle.f64_bits(L.wbo_perf_get_usage(u)) == oracle.f64_bits(reflib.ref_perf_get_usage(u)), u
    a = b = 0.0
    for i in range(2000):
        d = 10.0 * (0.4 + 0.3 * np.sin(i / 23.0)) + float(rng.random())
        a, b = L.wbo_perf_update(a, d, 10.666666666666666), reflib.ref_perf_update(b, d, 10.666666666666666)
        assert oracle.f64_bits(a) == oracle.f64_bits(b), i
    for rate in (8000, 11025, 22050, 32000, 44100, 48000, 88200, 96000, 176400, 192000):
        for size in range(4, 4097, 4):
            assert oracle.f64_bits(L.wbo_buffer_duration_ms(size, rate)) == oracle.f64_bits(reflib.ref_buffer_duration_ms(size, rate)), (size, rate)
