"""Quirk Q12 and the "wild" sessions (round 5).

Q12 — a NEGATIVE playback speed.  calc_resize_clip's stretch (clip_edit.h:59-67,110-118) sets
`new_speed = sample_count / (old_length + num_samples)`; shrinking a clip by more than its sample's stretched length makes the
denominator, and with it the speed, negative (Engine::add_audio_clip accepts one as well).  Sampler::stream then runs the
position backwards: `(uint32_t)ceil((count - offset) / speed)` (sampler.cpp:102-104) wraps to a huge count, all num_samples
frames are rendered, and as soon as the position falls to -1 `src_sample[ix]` (sampler.cpp:53-54) reads the heap IN FRONT of
the channel array — undefined behaviour; in the compiled reference the values change from run to run (found by the round-4
review's differential against the compiled engine: its only divergence class in 1441 random sessions).  Oracle and product
DEFINE a tap at a negative index as 0; positions, ix = trunc(x) and the negative fraction x - ix are the reference's.

The wild sessions (tests/fuzz_util.py wild_session / run_wild_script) are the corners of what the reference's API accepts, all
at once: samples of 5 frames, 8 kHz sources in a 192 kHz session, blocks of 100 / 440 / 1000 frames, stretch factors 0.01 / 8 /
negative, zero gains, sub-frame clips, every host operation between blocks.

CPU: the oracle's statement of Q12 against an independent numpy transcription; the product's host code + sequencer source
(tests/cpp/host_sim.cpp) against the oracle on the wild scripts.  GPU (-m gpu): the product engine against the oracle — master,
peaks and plan rows bit for bit."""
import os

import numpy as np
import pytest

import fuzz_util as FZ
import host_sim as HS
import oracle_ffi as O
from whitebox_amd import synth


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def plan_rows(plan):
    return [(b, t, bo, ns, O.f64_bits(off), O.f64_bits(spd), O.f32_bits(g), smp)
            for (b, t, bo, ns, na, smp, off, spd, g, fl) in plan]


def oracle_rows(e, block):
    return [(block, t, ds, min(ln, 0xFFFF), O.f64_bits(off), O.f64_bits(spd), O.f32_bits(g), smp)
            for (t, ds, ln, off, spd, g, smp) in e.seglog()]


# ----------------------------------------------------------------------------------------------------------------------
# Q12: the oracle's statement
# ----------------------------------------------------------------------------------------------------------------------
def linear_rows_numpy(data, norm, norm_t, pos, speed, n, gain):
    """sample_linear (sampler.cpp:34-59) transcribed independently, taps at negative indices = 0; `data` channel-planar with
    the 16 zero frames of padding behind it, `norm_t` the type the normaliser multiplies in (np.float32 / np.float64)"""
    out = np.zeros((len(data), n), np.float32)
    for c, src in enumerate(data):
        for j in range(n):
            x = np.float64(pos) + np.float64(j) * np.float64(speed)
            ix = int(np.trunc(x))
            fx = np.float32(x - np.float64(ix))
            a = np.float32(norm_t(norm) * norm_t(src[ix] if ix >= 0 else 0))
            b = np.float32(norm_t(norm) * norm_t(src[ix + 1] if ix + 1 >= 0 else 0))
            s = np.float32(a + np.float32(fx * np.float32(b - a)))
            out[c, j] = np.float32(s * np.float32(gain))
    return out


@pytest.mark.parametrize("fmt", ["f32", "i16", "i24", "i32"])
def test_oracle_negative_speed_reads_zero_in_front_of_the_clip(fmt):
    """a sampler reset to offset 6.5 at speed -0.37: the position passes 0 at frame 18 and -1 at frame 21; from there on the
    taps in front of the clip are 0, the tap at index 0 still sounds while ix = -1, and the offset keeps running down"""
    rng = np.random.default_rng(12)
    count = 40
    if fmt == "f32":
        data = [np.concatenate([rng.uniform(-1, 1, count).astype(np.float32), np.zeros(16, np.float32)]) for _ in range(2)]
        norm, norm_t = 1.0, np.float32
    elif fmt == "i16":
        data = [np.concatenate([rng.integers(-32768, 32767, count).astype(np.int16), np.zeros(16, np.int16)]) for _ in range(2)]
        norm, norm_t = np.float32(1.0 / 32767.0), np.float32
    else:
        top = (1 << 23) - 1 if fmt == "i24" else (1 << 31) - 1
        data = [np.concatenate([rng.integers(-top - 1, top, count).astype(np.int32), np.zeros(16, np.int32)]) for _ in range(2)]
        norm, norm_t = 1.0 / float(top), np.float64
    s = O.OracleSampler(fmt, 2, 48000, count, data)
    s.reset(6.5, -0.37, 48000.0)
    out = [np.zeros(64, np.float32) for _ in range(2)]
    s.stream(out, 64, 0, 0.75)
    want = linear_rows_numpy(data, norm, norm_t, 6.5, -0.37, 64, 0.75)
    assert np.array_equal(bits(np.stack(out)), bits(want))
    # frames whose ix <= -2 (x <= -2): both taps in front of the clip -> exactly +0.0
    x = 6.5 + np.arange(64) * -0.37
    assert np.all(bits(np.stack(out))[:, x <= -2.0] == 0) and np.any(np.stack(out)[:, (x > -2.0) & (x <= -1.0)] != 0)
    assert s.state.sample_offset == 6.5 + 64 * -0.37           # sampler.cpp:103,209: the offset runs on, below zero
    out2 = [np.zeros(64, np.float32) for _ in range(2)]
    s.stream(out2, 64, 0, 0.75)                            # ... and the next call is not "finished" (offset < count): all zeros
    assert not np.any(bits(np.stack(out2)))


def _neg_session(fmt="f32", block=512, rate=48000, src_rate=48000, n_blocks=10, channels=2):
    """track 0: a clip played BACKWARDS from 300 frames into its sample (speed -0.75: below zero in the second block);
    track 1: a clip whose right edge is dragged left past its sample's stretched length with stretch (speed turns negative);
    track 2: an ordinary clip beside them"""
    beat = rate * 60.0 / 120.0
    samples = [synth.SampleSpec(i, 2, src_rate, 700 if i < 2 else 8000, fmt, 0.2 if fmt == "f32" else 1.0) for i in range(3)]
    clips = [synth.ClipSpec(0, 0.0, n_blocks * block / beat, 300.0, -0.75, 0.8, sample=0),
             synth.ClipSpec(1, 0.0, n_blocks * block / beat, 0.0, 1.0, 1.0, sample=1),
             synth.ClipSpec(2, 0.0, n_blocks * block / beat, 5.0, 1.0, 0.5, sample=2)]
    return synth.SessionSpec("q12", 3, 0x0C12, samples, clips, [-6.0, -3.0, -9.0], [0.3, -0.4, 0.0], [False] * 3,
                             bpm=120.0, sample_rate=rate, block=block, channels=channels), n_blocks


def test_oracle_stretch_shrink_past_the_sample_turns_the_speed_negative():
    """calc_resize_clip, right edge, stretch (clip_edit.h:59-67): a 700-frame sample under a 10-block clip, the edge dragged
    4.5 blocks to the left: 700 / (700 - 2304) < 0; the list carries that speed and the next event starts the sampler with it"""
    spec, n_blocks = _neg_session()
    e = O.build_oracle_engine(spec)
    e.enable_seglog()
    unit = spec.block / 24000.0
    e.resize_clip(1, 0, -4.5 * unit, 0.0, 1.0 / 96.0, False, False, True)
    sp = e.clips(1)[0][3]
    assert sp == 700.0 / (700.0 - 4.5 * 512) and sp < 0
    e.play()
    zero_blocks = 0
    for b in range(5):                     # (the clip now ends 5.5 blocks in)
        e.process()
        rows = [r for r in e.seglog() if r[0] == 1]
        assert len(rows) == 1 and rows[0][4] == sp and rows[0][2] == spec.block       # every frame of every block is streamed
        assert rows[0][3] == b * spec.block * sp                                     # from offset 0 downwards
        zero_blocks += int(not np.any(bits(e.peaks()[1])))
    assert zero_blocks == 4 and np.any(e.peaks()[2] != 0)   # only block 0 touches index 0 (x in (-1, 0]); track 2 plays on
    e.close()


# ----------------------------------------------------------------------------------------------------------------------
# the product's host code + sequencer source on the CPU
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("masked", [0, 1, 4])
def test_host_sequencer_negative_speed_session(masked):
    spec, n_blocks = _neg_session()
    e = O.build_oracle_engine(spec)
    e.enable_seglog()
    sim = HS.build_sim_engine(spec, max_blocks=n_blocks, masked_rows=masked)
    unit = spec.block / 24000.0
    e.resize_clip(1, 0, -4.5 * unit, 0.0, 1.0 / 96.0, False, False, True)
    sim.resize_clip(sim.tracks[1], 0, -4.5 * unit, 0.0, 1.0 / 96.0, False, False, True)
    assert FZ.clip_rows(sim.clips(sim.tracks[1])) == FZ.clip_rows(e.clips(1))
    e.play()
    sim.play()
    rows = []
    for b in range(n_blocks):
        e.process()
        rows += oracle_rows(e, b)
    sim.render(n_blocks)
    assert plan_rows(sim.fetch_plan()) == rows
    e.close()
    sim.close()


WILD_CPU = range(int(os.environ.get("WBX_WILD_FROM", "0")), int(os.environ.get("WBX_WILD_TO", "120")))


@pytest.mark.parametrize("seed", WILD_CPU)
def test_host_sequencer_wild_scripts(seed):
    """the wild sessions with every host operation between blocks: clip lists after every operation, the stream calls of every
    block and the transport — product host code + sequencer source against the oracle"""
    spec, n_blocks = FZ.wild_session(seed)
    e = O.build_oracle_engine(spec)
    e.enable_seglog()
    sim = HS.build_sim_engine(spec, max_blocks=2, masked_rows=[0, 1, 4][seed % 3])

    def on_block(b, trail):
        e.process()
        sim.render(1)
        assert plan_rows(sim.fetch_plan()) == oracle_rows(e, 0), (seed, b, trail[-4:])
        ph, sp, _ = sim.transport()
        assert (O.f64_bits(ph), O.f64_bits(sp)) == (O.f64_bits(e.playhead), O.f64_bits(e.sample_position)), (seed, b, trail[-4:])

    FZ.run_wild_script(seed, spec, n_blocks, e, sim, on_block)
    e.close()
    sim.close()


@pytest.mark.parametrize("segments", [0, 3])
@pytest.mark.parametrize("seed", range(0, 60))
def test_host_sequencer_wild_sessions_batched(seed, segments):
    """the same sessions without edits, all blocks in one plan (steady runs, shared templates, the overflow pool) — one walk
    per track and cut along the time axis"""
    spec, n_blocks = FZ.wild_session(seed)
    e = O.build_oracle_engine(spec)
    e.enable_seglog()
    e.play()
    rows = []
    for b in range(n_blocks):
        e.process()
        rows += oracle_rows(e, b)
    sim = HS.build_sim_engine(spec, max_blocks=n_blocks, masked_rows=[4, 0, 1][seed % 3], segments=segments)
    sim.play()
    sim.render(n_blocks)
    assert plan_rows(sim.fetch_plan()) == rows
    ph, sp, _ = sim.transport()
    assert (O.f64_bits(ph), O.f64_bits(sp)) == (O.f64_bits(e.playhead), O.f64_bits(e.sample_position))
    e.close()
    sim.close()


@pytest.mark.parametrize("seed", range(1000, 1100))
def test_host_sequencer_wild_scripts_in_pieces(seed):
    """the host's operations between RENDERS of 1 … 16 blocks (steady runs, shared templates and the overflow pool across the
    operations): the stream calls of every piece against the oracle's block-by-block log"""
    spec, n_blocks = FZ.wild_session(seed)
    e = O.build_oracle_engine(spec)
    e.enable_seglog()
    sim = HS.build_sim_engine(spec, max_blocks=16, masked_rows=[0, 1, 4][seed % 3])

    def on_piece(b, trail, k):
        rows, q12 = [], 0
        for i in range(k):
            e.process()
            rows += oracle_rows(e, i)
            q12 += FZ.q12_calls(e.seglog())
        sim.render(k)
        assert plan_rows(sim.fetch_plan()) == rows, (seed, b, k, trail[-4:])
        return q12

    FZ.run_wild_script(seed, spec, n_blocks, e, sim, on_piece, pieces=True)
    ph, sp, _ = sim.transport()
    assert (O.f64_bits(ph), O.f64_bits(sp)) == (O.f64_bits(e.playhead), O.f64_bits(e.sample_position))
    e.close()
    sim.close()


# ----------------------------------------------------------------------------------------------------------------------
# the device
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("fmt,src_rate,block,channels", [("f32", 48000, 512, 2), ("f32", 44100, 512, 2), ("i16", 48000, 256, 2),
                                                          ("i24", 96000, 128, 2), ("i32", 44100, 512, 1), ("f32", 22050, 1000, 2)])
def test_negative_speed_on_the_device(fmt, src_rate, block, channels):
    """Q12 through the product: the backwards clip and the stretch-shrunk one, callback path (block by block) and one batch
    render — master, peaks and plan rows equal to the oracle's, the taps in front of the clips read 0"""
    import whitebox_amd as W
    from whitebox_amd.engine import build_engine
    spec, n_blocks = _neg_session(fmt, block, 48000, src_rate, 10, channels)
    unit = spec.block / 24000.0
    for batch in (False, True):
        e = O.build_oracle_engine(spec)
        e.enable_seglog()
        eng = build_engine(spec, max_blocks=n_blocks)
        e.resize_clip(1, 0, -4.5 * unit, 0.0, 1.0 / 96.0, False, False, True)
        eng.resize_clip(eng.tracks[1], 0, -4.5 * unit, 0.0, 1.0 / 96.0, False, False, True)
        assert FZ.clip_rows(eng.clips(eng.tracks[1])) == FZ.clip_rows(e.clips(1))
        e.play()
        eng.play()
        oms, opk, rows = [], [], []
        for b in range(n_blocks):
            om, _ = e.process()
            oms.append(om)
            opk.append(e.peaks())
            rows += oracle_rows(e, b)
        if batch:
            eng.render(n_blocks)
            m, pk, _ = eng.ctx.fetch(peaks=True)
            assert plan_rows(eng.fetch_plan()) == rows
            assert np.array_equal(bits(m), bits(np.stack(oms)))
            assert np.array_equal(pk, np.stack(opk)[..., :channels])
        else:
            out = W.AudioBuffer(spec.block, spec.channels)
            for b in range(n_blocks):
                eng.process(None, out, float(spec.sample_rate))
                assert np.array_equal(bits(np.stack(out.channel_buffers)), bits(oms[b])), b
                _, pk, _ = eng.ctx.fetch(peaks=True)
                assert np.array_equal(pk[0], opk[b][:, :channels]), b
                assert plan_rows(eng.fetch_plan()) == [(0,) + r[1:] for r in rows if r[0] == b], b
        e.close()
        eng.close()


WILD_GPU = range(int(os.environ.get("WBX_WILD_GPU_FROM", "0")), int(os.environ.get("WBX_WILD_GPU_TO", "60")))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", WILD_GPU)
def test_wild_scripts_on_the_device(seed):
    """the wild scripts through the product engine, one callback per block: master (the sessions have at most 5 tracks — one
    group, the reference's order), peaks, plan rows and transport bit for bit"""
    import whitebox_amd as W
    from whitebox_amd.engine import build_engine
    spec, n_blocks = FZ.wild_session(seed)
    e = O.build_oracle_engine(spec)
    e.enable_seglog()
    eng = build_engine(spec, max_blocks=2)
    out = W.AudioBuffer(spec.block, spec.channels)

    def on_block(b, trail):
        om, _ = e.process()
        eng.process(None, out, float(spec.sample_rate))
        assert plan_rows(eng.fetch_plan()) == oracle_rows(e, 0), (seed, b, trail[-4:])
        assert np.array_equal(bits(np.stack(out.channel_buffers)), bits(om[:spec.channels])), (seed, b, trail[-4:])
        _, pk, _ = eng.ctx.fetch(peaks=True)
        nt = len(eng.tracks)
        assert np.array_equal(pk[0][:nt], e.peaks()[:nt, :spec.channels]), (seed, b, trail[-4:])

    FZ.run_wild_script(seed, spec, n_blocks, e, eng, on_block)
    ph, sp, _ = eng.transport()
    assert (O.f64_bits(ph), O.f64_bits(sp)) == (O.f64_bits(e.playhead), O.f64_bits(e.sample_position))
    e.close()
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(0, 40))
def test_wild_sessions_batched_on_the_device(seed):
    """the wild sessions without edits as ONE device pass over all their blocks"""
    from whitebox_amd.engine import build_engine
    spec, n_blocks = FZ.wild_session(seed)
    e = O.build_oracle_engine(spec)
    e.enable_seglog()
    e.play()
    oms, opk, rows = [], [], []
    for b in range(n_blocks):
        om, _ = e.process()
        oms.append(om[:spec.channels])
        opk.append(e.peaks())
        rows += oracle_rows(e, b)
    eng = build_engine(spec, max_blocks=n_blocks)
    eng.play()
    eng.render(n_blocks)
    m, pk, _ = eng.ctx.fetch(peaks=True)
    assert plan_rows(eng.fetch_plan()) == rows
    assert np.array_equal(bits(m), bits(np.stack(oms)))
    assert np.array_equal(pk, np.stack(opk)[..., :spec.channels])
    e.close()
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("WBX_WILD_PIECES_FROM", "1000")), int(os.environ.get("WBX_WILD_PIECES_TO", "1040"))))
def test_wild_scripts_in_pieces_on_the_device(seed):
    """the wild scripts with the host's operations BETWEEN RENDERS of 1 … 16 blocks (from 8 blocks on the batch path: sequencer
    on the plan stream, pre-render pass, chained / grouped sums): every block of every piece against the oracle"""
    from whitebox_amd.engine import build_engine
    spec, n_blocks = FZ.wild_session(seed)
    e = O.build_oracle_engine(spec)
    e.enable_seglog()
    eng = build_engine(spec, max_blocks=16)

    def on_piece(b, trail, k):
        oms, opk, rows, q12 = [], [], [], 0
        for i in range(k):
            om, _ = e.process()
            oms.append(om[:spec.channels])
            opk.append(e.peaks())
            rows += oracle_rows(e, i)
            q12 += FZ.q12_calls(e.seglog())
        eng.render(k)
        m, pk, _ = eng.ctx.fetch(peaks=True)
        assert plan_rows(eng.fetch_plan()) == rows, (seed, b, k, trail[-4:])
        assert np.array_equal(bits(m), bits(np.stack(oms))), (seed, b, k, trail[-4:])
        nt = len(eng.tracks)
        assert np.array_equal(pk[:, :nt], np.stack(opk)[:, :nt, :spec.channels]), (seed, b, k, trail[-4:])
        return q12

    FZ.run_wild_script(seed, spec, n_blocks, e, eng, on_piece, pieces=True)
    ph, sp, _ = eng.transport()
    assert (O.f64_bits(ph), O.f64_bits(sp)) == (O.f64_bits(e.playhead), O.f64_bits(e.sample_position))
    e.close()
    eng.close()
