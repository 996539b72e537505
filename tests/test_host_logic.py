"""Host-side logic of the product (libwbx.so entry points that need no device): the clip placement
arithmetic of src/engine/clip_edit.h, against the committed golden vectors (outputs of the reference's
own header) and against the oracle on random inputs — fp64 bit patterns."""
import ctypes as C
import os

import numpy as np

import golden_util as G
import oracle_ffi as O
import whitebox_amd as W


def _product(vals, flags):
    L = W.lib()
    d = [C.c_double() for _ in range(4)]
    L.wbx_calc_resize_clip(*vals, *flags, *[C.byref(x) for x in d])
    got = [O.f64_bits(x.value) for x in d]
    L.wbx_calc_move_clip(vals[0], vals[1], vals[6], vals[9], C.byref(d[0]), C.byref(d[1]))
    got += [O.f64_bits(d[0].value), O.f64_bits(d[1].value),
            O.f64_bits(L.wbx_calc_clip_shift(vals[2], vals[6], vals[10], vals[4])),
            O.f64_bits(L.wbx_shift_clip_content(vals[2], vals[3], vals[4], vals[6], vals[10]))]
    return got


def test_clip_edit_arithmetic_matches_reference_golden():
    g = np.load(os.path.join(G.GOLDEN, "clip_edit.npz"))
    for k, want in zip(g["inputs"], g["outputs"]):
        assert _product([float(x) for x in k[:11]], [int(x) for x in k[11:]]) == [int(x) for x in want]


def test_clip_edit_arithmetic_matches_oracle_random(oracle):
    Lo = oracle.lib()
    rng = np.random.default_rng(77)
    d = [C.c_double() for _ in range(4)]
    for _ in range(2000):
        mn = float(rng.uniform(0, 200))
        vals = [mn, mn + float(rng.uniform(1e-3, 40)), float(rng.uniform(0, 1e6)), float(rng.uniform(0.1, 3.0)),
                float(rng.choice([22050, 44100, 48000, 96000, 192000])), float(rng.integers(1, 10_000_000)),
                float(rng.normal(0, 6)), float(rng.uniform(0, 2)), float(rng.uniform(1e-4, 1)), float(rng.uniform(0, 8)),
                60.0 / float(rng.uniform(30, 300))]
        flags = [int(rng.integers(0, 2)) for _ in range(4)]
        Lo.wbo_calc_resize_clip(*vals, *flags, *[C.byref(x) for x in d])
        want = [O.f64_bits(x.value) for x in d]
        Lo.wbo_calc_move_clip(vals[0], vals[1], vals[6], vals[9], C.byref(d[0]), C.byref(d[1]))
        want += [O.f64_bits(d[0].value), O.f64_bits(d[1].value),
                 O.f64_bits(Lo.wbo_calc_clip_shift(vals[2], vals[6], vals[10], vals[4])),
                 O.f64_bits(Lo.wbo_shift_clip_content(vals[2], vals[3], vals[4], vals[6], vals[10]))]
        assert _product(vals, flags) == want


def test_bench_algorithmic_bytes_match_survey_figures():
    """16.78 MB per 4096-track block at unity rate, 15.41 MB at 44.1 -> 48 kHz (clip reads only;
    bench.py adds the master write, the peaks and the 32 B of tables per track)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    extra = 512 * 2 * 4 + 4096 * 2 * 4 + 4096 * 32
    assert abs(b.algorithmic_bytes_per_block(4096, 48000) - (512 * 32768 + extra)) < 1e-6
    assert abs(b.algorithmic_bytes_per_block(4096, 44100) - (512 * 32768 * 0.91875 + extra)) < 1e-3
    assert abs(b.algorithmic_bytes_per_block(4096, 48000, fmt="i16") - (512 * 16384 + extra)) < 1e-6
    assert abs(512 * 32768 / 1e6 - 16.78) < 0.01 and abs(512 * 32768 * 0.91875 / 1e6 - 15.41) < 0.01


def test_perf_measurer_arithmetic_matches_reference_golden():
    """The product's load-figure arithmetic (wbx_calc_perf_update / wbx_calc_perf_usage / wbx_calc_buffer_period_ms: what
    wbx_engine_process feeds Engine::perf_measurer with, engine.cpp:52,1653) against the outputs of the reference's own
    PerformanceMeasurer (core/timing.h:54-67) and period helpers (engine/audio_io.h:187-195) — tests/golden/perf.npz, fp64 bit
    patterns; host-only entry points, no device"""
    g = np.load(os.path.join(G.GOLDEN, "perf.npz"))
    L = W.lib()
    u, d, t = (g[k].view(np.float64) for k in ("usage", "duration_ms", "period_ms"))
    upd = np.array([L.wbx_calc_perf_update(float(a), float(b), float(c)) for a, b, c in zip(u, d, t)])
    assert np.array_equal(upd.view(np.uint64), g["updated"])
    use = np.array([L.wbx_calc_perf_usage(float(a)) for a in np.concatenate([u, upd])])
    assert np.array_equal(use.view(np.uint64), g["clamped"])
    ms = np.array([L.wbx_calc_buffer_period_ms(int(b), int(r)) for b, r in g["pairs"]])
    assert np.array_equal(ms.view(np.uint64), g["buffer_ms"])
    cur, run = 0.0, []
    for x in g["run_durations"].view(np.float64):
        cur = L.wbx_calc_perf_update(cur, float(x), 10.666666666666666)
        run.append(cur)
    assert np.array_equal(np.array(run).view(np.uint64), g["run_usage"])


def test_perf_measurer_arithmetic_matches_oracle_random(oracle):
    Lo, L = oracle.lib(), W.lib()
    rng = np.random.default_rng(78)
    for _ in range(3000):
        u, d, t = float(rng.random() * 1.4 - 0.2), float(10.0 ** (rng.random() * 6 - 3)), float(rng.uniform(0.5, 50.0))
        assert O.f64_bits(L.wbx_calc_perf_update(u, d, t)) == O.f64_bits(Lo.wbo_perf_update(u, d, t))
        assert O.f64_bits(L.wbx_calc_perf_usage(u)) == O.f64_bits(Lo.wbo_perf_get_usage(u))
        b, r = int(rng.integers(1, 8193)) * 4, int(rng.choice([8000, 22050, 44100, 48000, 96000, 192000]))
        assert O.f64_bits(L.wbx_calc_buffer_period_ms(b, r)) == O.f64_bits(Lo.wbo_buffer_duration_ms(b, r))
