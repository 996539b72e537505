#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE'S OWN code (oracle/_ref/libwbref.so).

Run in the container that has /root/reference:   python oracle/gen_golden.py
The fixtures are data only — seeds/parameters of the synthetic inputs and the outputs the reference's
translation units produced for them; no reference source is stored.  Block sequencing for the session
fixtures comes from the oracle's restated sequencer (the reference's engine.cpp/track.cpp cannot be
built here); every sample value in the fixtures was computed by reference code."""
import ctypes as C
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_ffi as O          # noqa: E402
from refmix import RefMixer     # noqa: E402
from whitebox_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

SAMPLER_CASES = [(fmt, rate, speed) for fmt in ("f32", "i16", "i24", "i32")
                 for rate, speed in ((48000, 1.0), (44100, 1.0), (48000, 0.5), (48000, 1.75), (96000, 1.0))]

SESSIONS = {
    "c1": dict(n_tracks=8, clip_channels=1, unity_gain=True, seed=0x5EED0001),
    "c2": dict(n_tracks=48, seed=0x5EED0002),
    "c3": dict(n_tracks=40, src_rate=44100, seed=0x5EED0003),
    "c4": dict(n_tracks=64, n_buses=8, seed=0x5EED0004),
    "seek": dict(n_tracks=24, seek=True, seed=0x5EED0005),
    "seek441": dict(n_tracks=24, seek=True, src_rate=44100, seed=0x5EED0006),
    "hot": dict(n_tracks=16, amp=0.5, seed=0x5EED0007),
    "i16": dict(n_tracks=12, fmt="i16", seed=0x5EED0008),
    "i24_441": dict(n_tracks=12, fmt="i24", src_rate=44100, seed=0x5EED0009),
}
N_BLOCKS = 6


def sampler_ops(fmt, rate, speed):
    rng = np.random.default_rng(zlib.crc32(repr((fmt, rate, speed)).encode()))
    start = float(rng.integers(0, 40))
    ops = []
    for _ in range(8):
        n = int(rng.choice([512, 512, 1, 0, 37, 255, 300]))
        boff = int(rng.integers(0, 512 - n + 1)) if n < 512 else 0
        gain = float(np.float32(rng.choice([1.0, 0.5, 0.3333])))
        ops.append((n, boff, gain))
    return start, ops


def main():
    R = O.ref()
    if R is None:
        raise SystemExit("oracle/_ref is not built (needs /root/reference)")
    os.makedirs(OUT, exist_ok=True)

    # ---- scalars ---------------------------------------------------------------------------------
    pans = np.linspace(-1, 1, 257).astype(np.float32)
    pan_bits = np.zeros((5, len(pans), 2), np.uint32)
    for law in range(5):
        for i, p in enumerate(pans):
            l, r = C.c_float(), C.c_float()
            R.ref_pan_coefs(p, law, C.byref(l), C.byref(r))
            pan_bits[law, i] = (O.f32_bits(l.value), O.f32_bits(r.value))
    dbs = np.linspace(-80, 12, 369).astype(np.float32)
    db_bits = np.array([O.f32_bits(R.ref_db_to_linear(d)) for d in dbs], np.uint32)
    np.savez_compressed(os.path.join(OUT, "scalars.npz"), pans=pans, pan_bits=pan_bits, dbs=dbs, db_bits=db_bits)

    # ---- Sampler::stream -------------------------------------------------------------------------
    sam = {}
    for fmt, rate, speed in SAMPLER_CASES:
        count = 3000
        sess = synth.SessionSpec("s", 1, 0xABCD, [synth.SampleSpec(7, 2, rate, count, fmt, 0.7)], [], [0], [0], [False])
        data = sess.sample_data(0)
        ptrs = O.void_ptrs(data)
        start, ops = sampler_ops(fmt, rate, speed)
        ps, so = C.c_double(), C.c_double()
        R.ref_sampler_reset(C.byref(ps), C.byref(so), start, speed, float(rate), 48000.0)
        outs, offs = [], []
        for n, boff, gain in ops:
            b = [np.zeros(512, np.float32) for _ in range(2)]
            R.ref_sampler_stream(C.byref(ps), C.byref(so), O.FMT[fmt], 2, rate, count, C.cast(ptrs, O.c_voidpp), 2, n,
                                 boff, np.float32(gain), O.planar_ptrs(b))
            outs.append(np.stack(b))
            offs.append(so.value)
        key = f"{fmt}_{rate}_{speed}"
        sam[key + "_out"] = np.stack(outs)
        sam[key + "_off"] = np.array(offs, np.float64)
    np.savez_compressed(os.path.join(OUT, "sampler.npz"), **sam)

    # ---- whole blocks ----------------------------------------------------------------------------
    manifest = {}
    for name, kw in SESSIONS.items():
        kw2 = dict(kw)
        spec = synth.make_session(name, kw2.pop("n_tracks"), n_blocks=N_BLOCKS, **kw2)
        e = O.build_oracle_engine(spec)
        e.enable_seglog()
        rm = RefMixer(spec)
        e.play()
        masters, peaks, buses, gains, segs = [], [], [], [], []
        for b in range(N_BLOCKS):
            e.process()
            log = e.seglog()
            g = e.gains()
            m, bu, pk, ends = rm.block(log, g)
            masters.append(m)
            peaks.append(pk)
            gains.append(g)
            if bu is not None:
                buses.append(bu)
            for (t, ds, ln, off, spd, cg, smp), end in zip(log, ends):
                segs.append((b, t, ds, ln, O.f64_bits(off), O.f64_bits(spd), O.f32_bits(cg), smp, O.f64_bits(end)))
        e.close()
        arrs = dict(master=np.stack(masters), peaks=np.stack(peaks), gains=np.stack(gains).view(np.uint32),
                    segs=np.array(segs, np.uint64))
        if buses:
            arrs["buses"] = np.stack(buses)
        np.savez_compressed(os.path.join(OUT, f"session_{name}.npz"), **arrs)
        manifest[name] = kw
    with open(os.path.join(OUT, "sessions.json"), "w") as f:
        json.dump({"n_blocks": N_BLOCKS, "sessions": manifest}, f, indent=1, sort_keys=True)

    # ---- format conversion -----------------------------------------------------------------------
    rng = np.random.default_rng(5)
    src = [np.clip(rng.normal(0, 0.5, 600), -1, 1).astype(np.float32) for _ in range(2)]
    src[0][:6] = [1.0, -1.0, 0.0, -0.0, 0.9999999, -0.9999999]
    conv = {"src": np.stack(src)}
    for name, dt, width in (("i16", np.int16, 1), ("i24_x8", np.int32, 1), ("i32", np.int32, 1), ("f32", np.float32, 1),
                            ("i24", np.uint8, 3)):
        b = np.zeros(512 * 2 * width, dt)
        getattr(R, "ref_f32_to_" + name)(b.ctypes.data, O.planar_ptrs(src), 40, 512, 2)
        conv[name] = b
    np.savez_compressed(os.path.join(OUT, "conv.npz"), **conv)

    # ---- clip placement arithmetic (engine/clip_edit.h) -----------------------------------------------
    rng = np.random.default_rng(11)
    rows_in, rows_out = [], []
    d = [C.c_double() for _ in range(4)]
    for _ in range(400):
        mn = float(rng.uniform(0, 64)); ln = float(rng.uniform(0.01, 16))
        k = [mn, mn + ln, float(rng.uniform(0, 50000)), float(rng.choice([1.0, 0.5, 1.25, 0.91875])),
             float(rng.choice([44100, 48000, 96000])), float(rng.integers(1000, 2000000)), float(rng.uniform(-8, 8)),
             float(rng.uniform(0, 1)), float(rng.uniform(0.001, 0.5)), float(rng.uniform(0, 4)),
             60.0 / float(rng.uniform(60, 200))]
        flags = [int(rng.integers(0, 2)) for _ in range(4)]
        R.ref_calc_resize_clip(*k, *flags, *[C.byref(x) for x in d])
        res = [x.value for x in d]
        R.ref_calc_move_clip(k[0], k[1], k[6], k[9], C.byref(d[0]), C.byref(d[1]))
        res += [d[0].value, d[1].value, R.ref_calc_clip_shift(k[2], k[6], k[10], k[4]),
                R.ref_shift_clip_content(k[2], k[3], k[4], k[6], k[10])]
        rows_in.append(k + [float(f) for f in flags])
        rows_out.append(res)
    np.savez_compressed(os.path.join(OUT, "clip_edit.npz"), inputs=np.array(rows_in, np.float64),
                        outputs=np.array(rows_out, np.float64).view(np.uint64))
    print("golden vectors written to", OUT)



def mip_inputs():
    """(name, fmt, data) of the mip-map fixture: every storage format the summariser reads, counts around the level and
    pair boundaries, plateaus, extremes, float values outside [-1, 1], ±Inf and NaN"""
    rng = np.random.default_rng(0x3190)
    out = []
    for count in (65, 66, 130, 257, 1030, 4099, 9000):
        f = (rng.standard_normal(count) * 0.4).astype(np.float32)
        f[rng.integers(0, count, 6)] = [1.0, -1.0, 1.5, -2.25, 0.0, -0.0]
        out.append((f"f32_{count}", "f32", f))
        i16 = rng.integers(-32768, 32768, count).astype(np.int16)
        i16[rng.integers(0, count, 4)] = [32767, -32768, 0, -1]
        i16[10:40] = 1234                                  # a plateau: min / max order by first occurrence
        out.append((f"i16_{count}", "i16", i16))
        i32 = rng.integers(-2**31, 2**31, count).astype(np.int32)
        i32[rng.integers(0, count, 4)] = [2**31 - 1, -2**31, 0, -1]
        out.append((f"i32_{count}", "i32", i32))
    wild = (rng.standard_normal(1030) * 1e6).astype(np.float32)
    wild[[5, 77, 300]] = [np.inf, -np.inf, np.nan]
    out.append(("f32_wild_1030", "f32", wild))
    return out


def gen_mip():
    """tests/golden/mip.npz from the reference's own summarize_for_mipmaps_impl (oracle/_ref/libwbref_mip.so)."""
    if O.ref_mip() is None:
        raise SystemExit("oracle/_ref/libwbref_mip.so is not built (needs /root/reference)")
    arrs = {}
    for name, fmt, data in mip_inputs():
        arrs[f"{name}.in"] = data if fmt != "f32" else data.view(np.uint32)      # float inputs as bit patterns (NaN payloads)
        for q in (0, 1):
            for level in range(O.oracle_mip_levels(len(data))):
                arrs[f"{name}.q{q}.l{level}"] = O.ref_mip_level(fmt, data, level, q)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "mip.npz"), **arrs)
    print("mip.npz:", len(arrs), "arrays")



def vu_inputs():
    """(blocks [n_blocks][n] fp32, reset_every) of the VU fixture: ordinary levels, silence, all -0.0 (a muted track),
    NaN at the start / middle / end of a block and whole-NaN blocks (the running maximum restarts after a NaN,
    math::max(a, b) = b < a ? a : b), infinities, resets between blocks"""
    rng = np.random.default_rng(0x7005)
    out = []
    for n in (1, 7, 128, 512):
        for scale in (0.0, 1e-3, 0.5, 3.0):
            a = (rng.standard_normal((9, n)) * scale).astype(np.float32)
            out.append((a, 0))
        a = (rng.standard_normal((9, n)) * 0.5).astype(np.float32)
        a[1, 0] = np.nan; a[3, n // 2] = np.nan; a[5, n - 1] = np.nan; a[7, :] = np.nan
        out.append((a, 0)); out.append((a.copy(), 3))
        b = (rng.standard_normal((6, n)) * 0.25).astype(np.float32)
        b[2, :] = -0.0; b[4, n // 3] = np.inf; b[5, 0] = -np.inf
        out.append((b, 0)); out.append((b.copy(), 2))
    return out


def gen_vu():
    """tests/golden/vu.npz from the reference's own VUMeter (oracle/_ref/libwbref_vu.so)."""
    if O.ref_vu() is None:
        raise SystemExit("oracle/_ref/libwbref_vu.so is not built (needs /root/reference)")
    arrs = {}
    for i, (a, reset) in enumerate(vu_inputs()):
        arrs[f"c{i:02d}.in"] = a.view(np.uint32)
        arrs[f"c{i:02d}.reset"] = np.array(reset)
        arrs[f"c{i:02d}.levels"] = O.ref_vu_levels(a, reset).view(np.uint32)
    np.savez_compressed(os.path.join(OUT, "vu.npz"), **arrs)
    print("vu.npz:", len(arrs) // 3, "cases")


def perf_inputs():
    """(usage, duration_ms, period_ms) triples for PerformanceMeasurer::update / get_usage and (buffer_size, sample_rate) pairs
    for Engine::audio_buffer_duration_ms: running averages from idle to overload, sub-microsecond and multi-second blocks,
    usage outside [0, 1] and non-finite (the clamp's compares), every block size / rate the back ends open"""
    rng = np.random.default_rng(0x9E2F)
    u = np.concatenate([[0.0, 1.0, 0.25, -0.5, 2.0, np.inf, -np.inf, np.nan, 5e-324, 1 - 2 ** -53], rng.random(190) * 1.5 - 0.2])
    d = np.concatenate([[0.0, 10.0, 1e-4, 3.0, 40.0, 1.0, 1.0, 1.0, 5e-324, 10.666666666666666], 10.0 ** (rng.random(190) * 6 - 3)])
    t = np.concatenate([[10.0, 10.0, 10.666666666666666, 2.9, 5.3, 10.0, 10.0, 10.0, 10.0, 10.666666666666666],
                        rng.choice([1.3, 2.6, 2.9, 5.3, 5.8, 10.0, 10.6, 10.7, 21.3, 42.7], 190)])
    sizes = [32, 64, 96, 128, 192, 200, 256, 333, 416, 441, 448, 480, 512, 960, 1000, 1024, 1440, 1920, 2048, 4096, 32768]
    rates = [8000, 11025, 22050, 32000, 44100, 48000, 88200, 96000, 176400, 192000]
    pairs = np.array([(b, r) for b in sizes for r in rates], np.uint32)
    return u, d, t, pairs


def gen_perf():
    """tests/golden/perf.npz from the reference's own PerformanceMeasurer (core/timing.h) and period helpers (engine/audio_io.h),
    both header-only and compiled into oracle/_ref/libwbref.so"""
    R = O.ref()
    if R is None:
        raise SystemExit("oracle/_ref/libwbref.so is not built (needs /root/reference)")
    u, d, t, pairs = perf_inputs()
    upd = np.array([R.ref_perf_update(float(a), float(b), float(c)) for a, b, c in zip(u, d, t)])
    use = np.array([R.ref_perf_get_usage(float(a)) for a in np.concatenate([u, upd])])
    ms = np.array([R.ref_buffer_duration_ms(int(b), int(r)) for b, r in pairs])
    # a measurer followed over 400 blocks of a drifting load (the running average's accumulated rounding)
    rng = np.random.default_rng(0x9E30)
    durs = 10.0 * (0.3 + 0.25 * np.sin(np.arange(400) / 17.0) + 0.05 * rng.random(400))
    run, cur = [], 0.0
    for x in durs:
        cur = R.ref_perf_update(cur, float(x), 10.666666666666666)
        run.append(cur)
    np.savez_compressed(os.path.join(OUT, "perf.npz"), usage=u.view(np.uint64), duration_ms=d.view(np.uint64), period_ms=t.view(np.uint64),
                        updated=upd.view(np.uint64), clamped=use.view(np.uint64), pairs=pairs, buffer_ms=ms.view(np.uint64),
                        run_durations=durs.view(np.uint64), run_usage=np.array(run).view(np.uint64))
    print("perf.npz:", len(u), "updates,", len(pairs), "periods, a run of", len(run))


def ingest_inputs():
    """(name, interleaved [frames][channels]) of the clip-ingest fixture: every storage format load_file hands to
    deinterleave_samples (I16; I32 = 24- and 32-bit files; F32), 1-6 channels, lengths around the decoder's 1024-frame chunk,
    extreme integers and float specials (NaN payloads, -0.0, infinities must come through untouched)"""
    rng = np.random.default_rng(0x1A6E57)
    out = []
    for ch in (1, 2, 3, 6):
        for frames in (1, 5, 1023, 1024, 1025, 2049):
            if ch > 2 and frames not in (5, 1025):
                continue
            i16 = rng.integers(-32768, 32768, (frames, ch)).astype(np.int16)
            i16[0, 0], i16[-1, -1] = -32768, 32767
            out.append((f"i16_c{ch}_n{frames}", i16))
            i32 = rng.integers(-2**31, 2**31, (frames, ch)).astype(np.int32)
            i32[0, -1], i32[-1, 0] = 2**31 - 1, -2**31
            out.append((f"i32_c{ch}_n{frames}", i32))
            f = (rng.standard_normal((frames, ch)) * 0.5).astype(np.float32)
            sp = np.array([0x7FC00001, 0xFFC12345, 0x7F800000, 0xFF800000, 0x80000000, 0x00000001], np.uint32).view(np.float32)
            idx = rng.integers(0, frames, 6)
            f[idx, rng.integers(0, ch, 6)] = sp
            out.append((f"f32_c{ch}_n{frames}", f))
    return out


def gen_ingest():
    """tests/golden/ingest.npz from the reference's own deinterleave_samples<T> under load_file's loop
    (oracle/_ref/libwbref_deint.so)."""
    if O.ref_deint() is None:
        raise SystemExit("oracle/_ref/libwbref_deint.so is not built (needs /root/reference)")
    arrs = {}
    for name, a in ingest_inputs():
        planar = O.ref_deinterleave(a)                      # [ch] of frames + 16 (sample_padding) elements
        bits = {2: np.uint16, 4: np.uint32}[a.dtype.itemsize]
        arrs[f"{name}.in"] = a.view(bits)
        arrs[f"{name}.out"] = np.stack(planar).view(bits)
    np.savez_compressed(os.path.join(OUT, "ingest.npz"), **arrs)
    print("ingest.npz:", len(arrs) // 2, "cases")


def gen_sequencer():
    """tests/golden/sequencer.npz from the reference's own sequencer and block driver (oracle/_ref/wbref_engine:
    Track::process_event / Track::process / Engine::process and the session-building calls, cut out of the reference's sources
    where they lie and compiled unmodified — oracle/ref_engine_driver.cpp).  Per session: the script (JSON: operations with exact
    floats, clip audio as keys of whitebox_amd.synth's generator) and the driver's answer file byte for byte (per block: master,
    playhead, sample_position, every track's AudioEvent list, sampler state and VU level; per operation whether the reference
    took it; clip lists after edits)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ref_engine as R
    import seq_sessions as S
    if not R.available():
        raise SystemExit("oracle/_ref/wbref_engine is not built (needs /root/reference)")
    arrs, n = {}, 0
    for kind, want in (("static", 10), ("controls", 10), ("edits", 12), ("dense", 8)):
        got, seed = 0, 1000
        while got < want:
            seed += 1
            s = S.session_script(seed, kind)
            if s.block % 4 and got % 4:          # three in four sessions in block shapes the product takes (multiples of 4 frames)
                continue
            try:
                orc = R.run_oracle(s)
            except R.Wrapped:
                continue                         # the reference writes past its block buffer there: no answer to record
            raw, ref = R.run_reference(s, want_raw=True)
            assert R.compare(ref, orc, f"{kind} {seed}") is None
            name = f"{kind}_{seed}"
            arrs[f"{name}.script"] = np.frombuffer(R.script_to_json(s).encode(), np.uint8)
            arrs[f"{name}.answer"] = np.frombuffer(raw, np.uint8)
            got += 1
            n += 1
    np.savez_compressed(os.path.join(OUT, "sequencer.npz"), **arrs)
    print("sequencer.npz:", n, "sessions")


BASELINE_REF_CASES = [   # (name, make_session kwargs, blocks) — BASELINE.json's configs and their seek variants
    ("c1", dict(n_tracks=8, clip_channels=1, unity_gain=True, seed=0x5EED0001), 8),
    ("c2", dict(n_tracks=256, seed=0x5EED0002), 8),
    ("c3", dict(n_tracks=4096, src_rate=44100, seed=0x5EED0003), 4),
    ("c3seek", dict(n_tracks=512, src_rate=44100, seek=True, seed=0x5EED0013), 8),
    ("c2seek", dict(n_tracks=256, seek=True, seed=0x5EED0005), 8),
    ("c5", dict(n_tracks=32768, seed=0x5EED0006), 2),
]


def gen_baseline_ref():
    """tests/golden/baseline_ref.npz: BASELINE.json's configurations 1, 2, 3 and 5 (and seek variants) rendered by the reference's
    OWN Engine::process (oracle/_ref/wbref_engine) — per block the master, playhead and sample_position; at the last block every
    track's VUMeter::level and Sampler::sample_offset_.  The sessions are whitebox_amd.synth.make_session(**kwargs) on every side."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ref_engine as R
    from whitebox_amd import synth
    if not R.available():
        raise SystemExit("oracle/_ref/wbref_engine is not built (needs /root/reference)")
    arrs = {}
    for name, kw, nb in BASELINE_REF_CASES:
        kw = dict(kw)
        spec = synth.make_session(name, kw.pop("n_tracks"), n_blocks=nb, **kw)
        ref = R.run_reference(R.script_from_spec(spec, nb), timeout=300)
        run = [r for r in ref if r[0] == "run"][0][1]
        assert all(r[1] == 1 for r in ref if r[0] == "op")          # every clip was taken
        arrs[f"{name}.master"] = np.stack([b["master"] for b in run])                                   # [blocks][C][F] bit patterns
        arrs[f"{name}.transport"] = np.array([[b["playhead"], b["sample_position"]] for b in run], np.uint64)
        last = run[-1]["tracks"]
        arrs[f"{name}.level"] = np.array([t["level"] for t in last], np.uint32)
        arrs[f"{name}.offset"] = np.array([t["offset"] for t in last], np.uint64)
        arrs[f"{name}.events"] = np.array([sum(len(t["events"]) for t in b["tracks"]) for b in run], np.uint32)
    # configuration 4 (64 sub-buses: an extension, the reference has no bus type) as SURVEY A13 defines its oracle: the reference's
    # own functions composed — one reference Engine per bus (64 x Engine::process over the bus's 64 tracks), the bus sums added in
    # bus order with the reference's AudioBuffer::mix; recorded: the UN-clamped master, a CRC of every bus sum, two bus sums whole
    import zlib
    spec = synth.make_session("c4", 4096, n_buses=64, n_blocks=4, seed=0x5EED0004)
    ref = R.run_reference(R.script_from_bus_spec(spec, 4), timeout=300)
    assert all(r[1] == 1 for r in ref if r[0] == "op")
    rb = [r for r in ref if r[0] == "runbus"][0][1]
    assert max(b["peak"] for b in rb) < 1.0          # no engine's own clamp touched its bus sum
    arrs["c4.master_unclamped"] = np.stack([b["master"] for b in rb])
    arrs["c4.bus_crc"] = np.array([[zlib.crc32(b["buses"][k].tobytes()) for k in range(64)] for b in rb], np.uint32)
    arrs["c4.bus0"] = np.stack([b["buses"][0] for b in rb])
    arrs["c4.bus63"] = np.stack([b["buses"][63] for b in rb])
    arrs["c4.transport"] = np.array([[b["playhead"], b["sample_position"]] for b in rb], np.uint64)
    arrs["cases"] = np.frombuffer(json.dumps([[n, k, b] for n, k, b in BASELINE_REF_CASES]).encode(), np.uint8)
    np.savez_compressed(os.path.join(OUT, "baseline_ref.npz"), **arrs)
    print("baseline_ref.npz:", len(BASELINE_REF_CASES), "configurations")

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] in ("mip", "vu", "ingest", "sequencer", "baseline", "perf"):     # (the other fixtures are untouched)
        {"mip": gen_mip, "vu": gen_vu, "ingest": gen_ingest, "sequencer": gen_sequencer, "baseline": gen_baseline_ref, "perf": gen_perf}[sys.argv[1]]()
    else:
        main()
        gen_mip()
        gen_vu()
        gen_ingest()
        gen_sequencer()
        gen_baseline_ref()
