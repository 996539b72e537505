"""CPU-side coverage of the product's layer-2 HOST code and of the sequencer source the plan kernel runs
(whitebox_amd/csrc/wbx_host.h, wbx_seq.h, wbx_clip_edit.h compiled with g++ into tests/cpp/host_sim.cpp): seek /
sample-index math, transport, block-rate gains and clip edits bit-equal to the oracle on the fuzz seeds of the GPU
suite, the plan-template budget, and a ThreadSanitizer run of the UI-thread / audio-thread contract.
No device, no audio samples: per-sample work exists only in the HIP kernels."""
import os
import subprocess

import numpy as np
import pytest

import fuzz_util as FZ
import host_sim as HS
import oracle_ffi as O
from whitebox_amd import synth


def plan_rows(plan):
    return [(b, t, bo, ns, O.f64_bits(off), O.f64_bits(spd), O.f32_bits(g), smp)
            for (b, t, bo, ns, na, smp, off, spd, g, fl) in plan]


def oracle_rows(e, block):
    return [(block, t, ds, min(ln, 0xFFFF), O.f64_bits(off), O.f64_bits(spd), O.f32_bits(g), smp)
            for (t, ds, ln, off, spd, g, smp) in e.seglog()]


def check_session(spec, n_blocks, batch, masked=False, segments=0):
    e = O.build_oracle_engine(spec)
    e.enable_seglog()
    e.play()
    rows = []
    for b in range(n_blocks):
        e.process()
        rows += oracle_rows(e, b)
    sim = HS.build_sim_engine(spec, max_blocks=n_blocks, masked_rows=masked, segments=segments)
    sim.play()
    if batch:
        sim.render(n_blocks)
        got = plan_rows(sim.fetch_plan())
        if segments:
            SEG_STATS.append(sim.segment_stats())
    else:
        got = []
        for b in range(n_blocks):
            sim.render(1)
            got += [(b,) + r[1:] for r in plan_rows(sim.fetch_plan())]
    assert got == rows
    ph, sp, _ = sim.transport()
    assert (O.f64_bits(ph), O.f64_bits(sp)) == (O.f64_bits(e.playhead), O.f64_bits(e.sample_position))
    # fl(volume * pan_c) per track (track.cpp:728-731), as bit patterns
    assert np.array_equal(sim.gains().view(np.uint32), e.gains().view(np.uint32))
    e.close()
    sim.close()


@pytest.mark.parametrize("masked", [0, 1, 4])
@pytest.mark.parametrize("seed", range(0, 160))
def test_host_sequencer_random_sessions_batched(seed, masked):
    """the 160 fuzz sessions of the GPU suite, all blocks in one plan (steady runs, templates, overflow pool); `masked`:
    planned for a mix instance that renders clip boundaries in its hot loop (masked rows, ROW_PAIRs)"""
    spec, n_blocks = FZ.random_session(seed)
    check_session(spec, n_blocks, batch=True, masked=masked)


SEG_STATS = []   # segment_stats() of every session planned by segments in this process


@pytest.mark.parametrize("masked,segments", [(0, 2), (1, 3), (4, 5), (1, 1)])
@pytest.mark.parametrize("seed", range(0, 160))
def test_host_sequencer_random_sessions_by_segments(seed, masked, segments):
    """... the same sessions with the sequencer cut along the time axis (wbx_seq.h plan_segment / plan_check_seams /
    plan_redo_track: what plan_seg_kernel runs on the device), segments of 1 / 2 / 3 / 5 blocks, the lanes in an order nothing may
    rely on: the stream-call log, the transport and the state the next render starts from are the one-walk plan's — whether a
    seam's guess held or the rest of the track was planned again."""
    spec, n_blocks = FZ.random_session(seed)
    check_session(spec, n_blocks, batch=True, masked=masked, segments=segments)


@pytest.mark.parametrize("seed", range(0, 40))
def test_host_sequencer_segments_then_more_renders(seed):
    """a segmented render hands the NEXT render the right state: the session in three batches (segmented, block by block,
    segmented again) against the oracle's block-by-block log"""
    spec, n_blocks = FZ.random_session(seed + 100000)
    e = O.build_oracle_engine(spec)
    e.enable_seglog()
    e.play()
    rows = []
    for b in range(n_blocks):
        e.process()
        rows += oracle_rows(e, b)
    sim = HS.build_sim_engine(spec, max_blocks=n_blocks, masked_rows=1, segments=2)
    sim.play()
    cuts = [0, n_blocks // 3, n_blocks // 3 + 1, n_blocks]
    got = []
    for lo, hi in zip(cuts, cuts[1:]):
        if hi > lo:
            sim.render(hi - lo)
            got += [(lo + r[0],) + r[1:] for r in plan_rows(sim.fetch_plan())]
    assert got == rows
    e.close()
    sim.close()


def test_segment_guesses_mostly_hold():
    """The point of the segmented plan is that a seam's guess nearly always holds (a miss is planned again in one walk: right,
    but slow).  Over sessions of back-to-back clips — what the plan is for — no seam may miss; over the fuzz sessions above
    (overlapping clips, edits, sub-block clips) most hold."""
    beat_frames = 48000 * 60.0 / 120.0
    for clip_blocks, n_blocks, seg in ((5.3, 96, 8), (1.3, 64, 4), (0.8, 48, 6), (20.0, 128, 16)):
        samples, clips = [], []
        n_tracks = 24
        for t in range(n_tracks):
            samples.append(synth.SampleSpec(seed_track=t, channels=2, rate=[48000, 44100][t % 2], frames=int((n_blocks + 2) * 512 * 1.2), fmt="f32", amp=0.02))
            L = clip_blocks * 512
            pos = -((t * 37) % 101) / 101.0 * L
            k = 0
            while pos < (n_blocks + 1) * 512:
                a, b = max(pos, 0.0), pos + L * (0.7 if (t + k) % 4 == 0 else 1.0)   # some gaps, mostly back to back
                if b > a:
                    clips.append(synth.ClipSpec(t, a / beat_frames, b / beat_frames, start_offset=a * 0.9, speed=1.0, gain=1.0, sample=t))
                pos += L
                k += 1
        spec = synth.SessionSpec(name="segs", n_tracks=n_tracks, seed=1, samples=samples, clips=clips, volumes_db=[0.0] * n_tracks,
                                 pans=[0.0] * n_tracks, mutes=[False] * n_tracks, block=512, channels=2)
        del SEG_STATS[:]
        check_session(spec, n_blocks, batch=True, masked=1, segments=seg)
        renders, tracks_redone, segs_redone, lanes = SEG_STATS[-1]
        assert renders == 1 and lanes == n_tracks * ((n_blocks + seg - 1) // seg - 1)
        assert tracks_redone == 0 and segs_redone == 0, (clip_blocks, SEG_STATS[-1])


@pytest.mark.parametrize("masked", [0, 1, 4])
@pytest.mark.parametrize("seed", range(0, 40))
def test_host_sequencer_random_sessions_block_by_block(seed, masked):
    """the second generator's sessions one block per plan (the audio-callback shape)"""
    spec, n_blocks = FZ.random_session(seed + 100000)
    check_session(spec, n_blocks, batch=False, masked=masked)


def test_clip_boundaries_stay_out_of_the_pre_render_queue():
    """An fp32 session cut into back-to-back clips (one clip ends and the next starts inside a block every few blocks):
    planned for a masked-row mix instance, every boundary block is a ROW_PAIR of two single-segment templates — nothing
    is queued for the pre-render pass, no overflow-pool chunk is taken — and the stream-call log still equals the
    oracle's.  Planned the old way, the same blocks are all queued."""
    n_tracks, K, L = 12, 40, 3.3 * 512
    spec = synth.make_session("cut", n_tracks, src_rate=44100, n_blocks=K, seed=0x0A74)
    beat = 24000.0
    clips = []
    for t in range(n_tracks):
        pos = -((t * 37) % 512) / 512.0 * L
        while pos < (K + 1) * 512:
            a, b = max(pos, 0.0), pos + L
            clips.append(synth.ClipSpec(t, a / beat, b / beat, start_offset=float(int(a * 0.91875)), speed=1.0, gain=1.0))
            pos = b
    spec.clips = clips
    e = O.build_oracle_engine(spec)
    e.enable_seglog()
    e.play()
    rows = []
    for b in range(K):
        e.process()
        rows += oracle_rows(e, b)
    e.close()
    stats = {}
    for masked in (False, True):
        sim = HS.build_sim_engine(spec, max_blocks=K, masked_rows=masked)
        sim.play()
        sim.render(K)
        assert plan_rows(sim.fetch_plan()) == rows, masked
        pc = sim.plan_counters()
        assert pc[1] == 0
        fl, kd = sim.row_kinds(K * n_tracks)
        stats[masked] = (pc[0], pc[2], sum(1 for f in fl if f & 4))
        sim.close()
    pool0, gen0, pairs0 = stats[False]
    pool1, gen1, pairs1 = stats[True]
    assert gen0 > n_tracks * 8 and pool0 == gen0 and pairs0 == 0     # old way: every boundary block pre-rendered
    assert gen1 == 0 and pool1 == 0 and pairs1 == gen0               # masked rows: all of them ROW_PAIRs in the hot loop


@pytest.mark.parametrize("seed", range(0, 80))
def test_host_sequencer_everything_family_sessions(seed):
    """the generator for the everything family (every storage format, speeds below and above 1), planned at level 4:
    every one- or two-call block a masked row / ROW_PAIR"""
    spec, n_blocks = FZ.random_masked_session(seed, everything=True)
    check_session(spec, n_blocks, batch=bool(seed & 1), masked=4)


@pytest.mark.parametrize("seed", range(40))
def test_host_sequencer_masked_row_sessions(seed):
    """the third generator: sessions the masked-row path takes (fp32, unity / window speeds, 512- and 1024-frame blocks)"""
    spec, n_blocks = FZ.random_masked_session(seed)
    check_session(spec, n_blocks, batch=bool(seed & 1), masked=True)


@pytest.mark.parametrize("seed", range(0, 60))
def test_host_sequencer_masked_integer_unity_sessions(seed):
    """sessions of integer-PCM clips at the session rate, planned at masked-row level 2: their clip boundaries are
    partial KIND_UNITY_I16 / KIND_UNITY_I32 records (ROW_PAIRs) for the hot loop, and the stream-call log is the oracle's"""
    spec, n_blocks = FZ.random_masked_session(seed, integer_unity=True)
    check_session(spec, n_blocks, batch=bool(seed & 1), masked=2)
    if seed < 8:   # and the boundary blocks really stay out of the pre-render queue (queued: blocks of 3+ stream calls only)
        stats = {}
        for level in (0, 2):
            sim = HS.build_sim_engine(spec, max_blocks=n_blocks, masked_rows=level)
            sim.play()
            sim.render(n_blocks)
            stats[level] = sim.plan_counters()
            sim.close()
        assert stats[2][2] <= stats[0][2]


@pytest.mark.parametrize("seed", range(0, 60))
def test_host_sequencer_masked_16bit_sessions(seed):
    """sessions of 16-bit PCM only, resampled at speeds up to 0.999 or not at all, planned at masked-row level 3 (the lean
    16-bit family of mix_kernel): partial KIND_WINDOW_I16 / KIND_UNITY_I16 records, the oracle's stream-call log"""
    spec, n_blocks = FZ.random_masked_session(seed, lean16=True)
    check_session(spec, n_blocks, batch=bool(seed & 1), masked=3)


@pytest.mark.parametrize("name,kw", [("c1", dict(n_tracks=8, channels_src=1)), ("seek", dict(n_tracks=24, seek=True)),
                                     ("seek441", dict(n_tracks=24, seek=True, src_rate=44100)),
                                     ("d96", dict(n_tracks=8, src_rate=96000)), ("long", dict(n_tracks=5, src_rate=44100))])
def test_host_sequencer_baseline_shapes(name, kw):
    kw = dict(kw)
    kw.pop("channels_src", None)
    n_blocks = 300 if name == "long" else 8
    spec = synth.make_session(name, n_blocks=n_blocks, seed=0x5EED0000 + len(name), **kw)
    check_session(spec, n_blocks, batch=True)


@pytest.mark.parametrize("seed", range(2024, 2064))
def test_host_clip_edits_match_oracle(seed):
    """random edit scripts (add with overlap / move / resize / delete / region delete / gain) between blocks: the clip
    lists and the plan of the following block stay bit-equal to the oracle's"""
    spec = FZ.edit_session_spec(seed)
    e = O.build_oracle_engine(spec)
    sim = HS.build_sim_engine(spec, max_blocks=2, masked_rows=bool(seed & 1))
    e.enable_seglog()

    def on_block(step, op):
        e.process()
        sim.render(1)
        assert plan_rows(sim.fetch_plan()) == oracle_rows(e, 0), (seed, step, op)

    FZ.run_edit_script(seed, spec, e, sim, on_block)
    ph, sp, _ = sim.transport()
    assert (O.f64_bits(ph), O.f64_bits(sp)) == (O.f64_bits(e.playhead), O.f64_bits(e.sample_position))
    e.close()
    sim.close()


def test_clip_outlasting_its_audio_shares_one_template():
    """A clip whose timeline region is far longer than its sample makes one zero-length 'finished' stream call per
    block (sampler.cpp:99-100) for the rest of the region.  Those calls share ONE plan template: a 256-block render of
    such tracks stays inside the template budget (it used to take one template per block and overflow)."""
    n_tracks, K = 3, 256
    spec = synth.make_session("outlast", n_tracks, n_blocks=4, seed=0x0A71)
    for s in spec.samples:
        s.frames = 700                    # ~1.4 blocks of audio
    for c in spec.clips:
        c.max_beat = c.min_beat + (K + 8) * 512 / 24000.0
    e = O.build_oracle_engine(spec)
    e.enable_seglog()
    e.play()
    rows = []
    for b in range(K):
        e.process()
        rows += oracle_rows(e, b)
    e.close()
    sim = HS.build_sim_engine(spec, max_blocks=K)
    sim.play()
    sim.render(K)
    assert plan_rows(sim.fetch_plan()) == rows          # every finished call is still in the plan
    pc = sim.plan_counters()
    assert pc[1] == 0, f"plan status bits {pc[1]}"
    assert pc[3] <= 32 * n_tracks, f"{pc[3]} templates for {n_tracks} tracks"   # one reservation (32 at K >= 64) per track
    assert sim.template_capacity() < K * n_tracks       # the budget really is smaller than one template per block
    sim.close()


def test_crawling_clip_gets_a_template_per_block():
    """(count - offset) / speed >= 2^32 (a clip slowed down a million-fold) leaves the shared-template path: the host
    sizes the template array for one template per block instead of reporting an overflow"""
    K = 64
    spec = synth.make_session("crawl", 2, n_blocks=4, seed=0x0A72)
    for c in spec.clips:
        c.speed = 1e-7
        c.max_beat = c.min_beat + (K + 8) * 512 / 24000.0
    e = O.build_oracle_engine(spec)
    e.enable_seglog()
    e.play()
    rows = []
    for b in range(K):
        e.process()
        rows += oracle_rows(e, b)
    e.close()
    sim = HS.build_sim_engine(spec, max_blocks=K)
    sim.play()
    sim.render(K)
    assert sim.plan_counters()[1] == 0
    assert plan_rows(sim.fetch_plan()) == rows
    sim.close()


def test_parameter_ring_is_block_rate_and_ordered():
    """Track::set_volume / set_pan / set_mute reach the audio side at the next block, in order (the last value wins),
    and the drain counters say how many messages each block took"""
    spec = synth.make_session("ring", 4, n_blocks=4, seed=0x0A73)
    e = O.build_oracle_engine(spec)
    sim = HS.build_sim_engine(spec, max_blocks=1)
    e.play()
    sim.play()
    e.process()
    sim.render(1)
    _, d0 = sim.thread_stats()
    assert all(x in (5, 6) for x in d0)   # 3 from Track::Track + set_volume + set_pan (+ set_mute)
    for k in range(20):
        for t in range(4):
            e.set_volume(t, -float(k + t))
            sim.tracks[t].set_volume(-float(k + t))
        e.set_pan(1, 0.01 * k)
        sim.tracks[1].set_pan(0.01 * k)
    e.set_mute(2, True)
    sim.tracks[2].set_mute(True)
    e.process()
    sim.render(1)
    assert np.array_equal(sim.gains().view(np.uint32), e.gains().view(np.uint32))
    _, d1 = sim.thread_stats()
    assert [b - a for a, b in zip(d0, d1)] == [20, 40, 21, 20]
    e.close()
    sim.close()


def test_two_thread_contract_under_thread_sanitizer():
    """UI thread (lock-free parameter rings + locked clip edits + tempo stores) against the audio thread's render,
    3000 blocks, built with -fsanitize=thread: no data race, every message drained, the last value of each fader wins."""
    exe = HS.build_tsan()
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "host_tsan ok" in r.stdout
    assert "WARNING: ThreadSanitizer" not in r.stderr


def replay_on_host_sim(name, s, want):
    """one script (tests/ref_engine.Script) through the product's host code + sequencer source and through the oracle: playhead /
    sample_position and the clip lists against `want` (the reference's answers, recorded or live), the stream calls of every block
    against the oracle replaying the same script.  -> blocks compared"""
    blocks = 0
    e = O.OracleEngine(s.channels, s.block, s.rate)
    e.enable_seglog()
    sim = HS.HostSimEngine(8, s.block, s.rate, s.channels, 1)
    sim.set_masked_rows(4)
    wi = 0
    for o in s.ops:
        rec = want[wi]
        wi += 1
        k = o[0]
        if k == "run":
            for br in rec[1]:
                e.process()
                sim.render(1)
                assert [(0,) + r[1:] for r in plan_rows(sim.fetch_plan())] == oracle_rows(e, 0), (name, br["block"])
                ph, sp, _ = sim.transport()
                assert (O.f64_bits(ph), O.f64_bits(sp)) == (br["playhead"], br["sample_position"]), (name, br["block"])
                blocks += 1
            continue
        if k == "query":
            continue
        if k == "clips":
            for t, tr in enumerate(sim.tracks):
                assert FZ.clip_rows(sim.clips(tr)) == rec[1][t], (name, t)
            continue
        if rec[1] != 1:
            continue
        if k == "bpm": e.set_bpm(o[1]); sim.set_bpm(o[1])
        elif k == "seek": e.set_playhead(o[1]); sim.set_playhead_position(o[1])
        elif k == "play": e.play(); sim.play()
        elif k == "stop": e.stop(); sim.stop()
        elif k in ("sample", "synth"):
            fmt, ch, rate, frames, data = s.samples[o[1]][:5]
            e.add_sample(fmt, ch, rate, frames, data); sim.add_sample_meta(fmt, ch, rate, frames)
        elif k == "track": e.add_track(); sim.add_track("t")
        elif k == "vol": e.set_volume(o[1], o[2]); sim.tracks[o[1]].set_volume(o[2])
        elif k == "pan": e.set_pan(o[1], o[2]); sim.tracks[o[1]].set_pan(o[2])
        elif k == "mute": e.set_mute(o[1], o[2]); sim.tracks[o[1]].set_mute(o[2])
        elif k == "clip":
            _, t, mn, mx, so, si, sp, g = o
            e.add_audio_clip(t, mn, mx, so, si, sp, g); sim.add_audio_clip(sim.tracks[t], "c", mn, mx, so, si, sp, g)
        elif k == "delclip": e.delete_clip(o[1], o[2]); sim.delete_clip(sim.tracks[o[1]], o[2])
        elif k == "gain": e.set_clip_gain(o[1], o[2], o[3]); sim.set_clip_gain(sim.tracks[o[1]], o[2], o[3])
        elif k == "move": e.move_clip(o[1], o[2], o[3]); sim.move_clip(sim.tracks[o[1]], o[2], o[3])
        elif k == "deltrack": e.delete_track(o[1]); sim.delete_track(o[1])
        elif k == "movetrack": e.move_track(o[1], o[2]); sim.move_track(o[1], o[2])
        elif k == "solo": e.solo_track(o[1]); sim.solo_track(o[1])
        else: assert k == "cfg", k
    e.close()
    sim.close()
    return blocks


def test_host_sequencer_on_the_reference_recordings():
    """tests/golden/sequencer.npz — what the reference's own Track::process_event / Track::process / Engine::process answered to
    40 session scripts (oracle/_ref/wbref_engine, oracle/gen_golden.py sequencer) — replayed through the product's host code +
    sequencer source: playhead / sample_position and the clip lists against the RECORDING, the stream calls of every block
    against the oracle replaying the same script (which test_sequencer_golden holds to the recording event for event)."""
    from test_oracle_golden import sequencer_golden_cases
    n = blocks = 0
    for name, s, want in sequencer_golden_cases():
        if s.block % 4 or any(o[0] == "rate" for o in s.ops):     # (the host sim has no reconfiguration call)
            continue
        blocks += replay_on_host_sim(name, s, want)
        n += 1
    assert n >= 24 and blocks > 300, (n, blocks)


@pytest.mark.ref
@pytest.mark.parametrize("kind", ["static", "controls", "edits", "dense", "wild"])
def test_host_sequencer_against_the_live_reference(kind):
    """the same against the reference executable itself (this container): fresh seeds of tests/seq_sessions.py.  WBX_REFSEQ_SEEDS
    widens the range."""
    import ref_engine as R
    import seq_sessions as S
    if not R.available():
        pytest.skip("oracle/_ref/wbref_engine not built (no /root/reference here)")
    n = int(os.environ.get("WBX_REFSEQ_SEEDS", "60"))
    done = blocks = 0
    first = int(os.environ.get("WBX_REFSEQ_FROM", "7000"))      # (soak runs: a seed range no earlier run has seen)
    for seed in range(first, first + n):
        s = S.session_script(seed, kind)
        if s.block % 4 or any(o[0] == "rate" for o in s.ops):
            continue
        try:
            R.run_oracle(s)
        except R.Wrapped:
            continue
        blocks += replay_on_host_sim(f"{kind} seed {seed}", s, R.run_reference(s))
        done += 1
    assert done >= n // 4, (done, n)
