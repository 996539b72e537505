"""The audio callback as ONE launch (wbx_callback.h: sequencer prologue + mix + spread sum + completion word), the path
wbx_engine_process takes for 512-frame stereo and 1024-frame mono blocks.  Reference call site: one Engine::process per device
period (audio_io_pulseaudio.cpp:396-466).  Against the oracle — per-track peaks and stream calls bit for bit, the master bit for
bit when one group holds a whole member list (the reference's order), within 1e-6 RMS otherwise — and against the batch render
of the same engine configuration (bit-identical: same functions, same order of additions)."""
import numpy as np
import pytest

import oracle_ffi as O
import whitebox_amd as W
from whitebox_amd import synth
from whitebox_amd.engine import build_engine

pytestmark = pytest.mark.gpu
RMS_TOL = 1e-6


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def plan_rows(plan):
    return [(b, t, bo, ns, O.f64_bits(off), O.f64_bits(spd), O.f32_bits(g), smp)
            for (b, t, bo, ns, na, smp, off, spd, g, fl) in plan]


def oracle_rows(e, block):
    return [(block, t, ds, min(ln, 0xFFFF), O.f64_bits(off), O.f64_bits(spd), O.f32_bits(g), smp)
            for (t, ds, ln, off, spd, g, smp) in e.seglog()]


def callback_group(n):       # what a max_blocks = 1 engine picks (wbx_runtime.hip build_routing)
    return 64 if n <= 16 else 1 if n <= 64 else 4 if n <= 256 else 8 if n <= 512 else 16


return_names = []      # the kernel names of the last run_callback call


def run_callback(spec, K, edits=None, check_plan=True, every_block_one_launch=True):
    """K blocks through Engine::process against the oracle block by block; `edits`: {block: fn(oracle, engine)} applied before it"""
    e = O.build_oracle_engine(spec)
    e.enable_seglog()
    eng = build_engine(spec, max_blocks=1)
    out = W.AudioBuffer(spec.block, spec.channels)
    e.play()
    eng.play()
    del return_names[:]
    masters, names = [], []
    one_group = spec.n_tracks <= 64 and not spec.n_buses      # one group, or one track per group: the reference's order
    for b in range(K):
        if edits and b in edits:
            edits[b](e, eng)
        om, _ = e.process()
        eng.process(None, out, float(spec.sample_rate))
        m = np.stack(out.channel_buffers)
        names.append(eng.ctx.kernel_name())
        if one_group:
            assert np.array_equal(bits(m), bits(om)), b
        else:
            d = m.astype(np.float64) - om.astype(np.float64)
            assert float(np.sqrt(np.mean(d * d))) <= RMS_TOL, b
        _, pk, _ = eng.ctx.fetch(peaks=True)
        # (by value: a muted track's peak is -0.0 in the reference — math::abs(-0.0f) is -0.0f and math::max keeps it — and +0.0
        #  here, where magnitudes are compared as unsigned integers)
        assert np.array_equal(pk[0], e.peaks()[:, :spec.channels]), b
        if check_plan:
            assert plan_rows(eng.fetch_plan()) == oracle_rows(e, 0), b
        masters.append(m.copy())
    ph, sp, _ = eng.transport()
    assert (O.f64_bits(ph), O.f64_bits(sp)) == (O.f64_bits(e.playhead), O.f64_bits(e.sample_position))
    e.close()
    eng.close()
    # (a block that queued records for the pre-render pass was repeated through the three kernels: its name is a mix_kernel's)
    one = [n.startswith("wbx::callback_kernel<") for n in names]
    assert all(one) or not every_block_one_launch, names
    return_names.extend(names)
    return np.stack(masters)


@pytest.mark.parametrize("n_tracks", [1, 8, 16, 17, 64, 65, 200, 300, 600, 1025, 4096])
@pytest.mark.parametrize("kw", [dict(src_rate=44100), dict(src_rate=44100, fmt="i16"), dict(src_rate=44100, fmt="i24"), dict()],
                         ids=["f32_441", "i16_441", "i24_441", "f32_unity"])
def test_one_launch_callback_equals_the_oracle_and_the_batch_render(n_tracks, kw):
    """every family's callback instance (0: fp32, 2: 16-bit resampled, 1 / 3: 24-bit resampled), one group / a few groups / one
    workgroup per CU; then the same blocks as ONE batch render with the callback's track groups: bit-identical"""
    K = 5
    spec = synth.make_session("cb", n_tracks, n_blocks=K, seed=0xCB00 + n_tracks, seek=n_tracks <= 200, **kw)
    if kw.get("fmt"):      # integer clips are full scale: the session level (master peak ~0.4) goes into the faders, as in bench.py
        spec.volumes_db = [v + 20.0 * float(np.log10(0.25 / np.sqrt(n_tracks))) for v in spec.volumes_db]
    got = run_callback(spec, K, check_plan=n_tracks <= 1025, every_block_one_launch=n_tracks > 200)
    eng = build_engine(spec, max_blocks=K, group_size=callback_group(n_tracks))
    eng.play()
    eng.render(K)
    batch, _, _ = eng.ctx.fetch()
    eng.close()
    assert np.array_equal(bits(got), bits(batch))


def test_one_launch_callback_more_groups_than_the_device_has_cus():
    """8192 tracks = 512 workgroups: more than one per CU, so no ticket barrier — the workgroup that sees the full count adds
    all group sums"""
    spec = synth.make_session("cb8k", 8192, src_rate=44100, n_blocks=3, seed=0xCB8)
    run_callback(spec, 3, check_plan=False)


@pytest.mark.parametrize("n_tracks,n_buses", [(48, 3), (256, 8), (4096, 64)])
def test_one_launch_callback_with_sub_buses(n_tracks, n_buses):
    """bus sums (bit-exact when a bus is one group) and master through the spread sum's sub-bus branch"""
    K = 4
    spec = synth.make_session("cbbus", n_tracks, n_blocks=K, seed=0xCBB, n_buses=n_buses)
    e = O.build_oracle_engine(spec)
    eng = build_engine(spec, max_blocks=1)
    out = W.AudioBuffer(spec.block, spec.channels)
    e.play()
    eng.play()
    per_bus = n_tracks // n_buses
    for b in range(K):
        om, obus = e.process(want_buses=True)
        eng.process(None, out, 48000.0)
        assert eng.ctx.kernel_name().startswith("wbx::callback_kernel<")
        m = np.stack(out.channel_buffers)
        _, pk, bus = eng.ctx.fetch(peaks=True, buses=True)
        assert np.array_equal(pk[0], e.peaks())
        if per_bus <= callback_group(n_tracks) or callback_group(n_tracks) == 1:      # every bus is one group, or every track: the reference's order
            assert np.array_equal(bits(bus[0]), bits(obus)) and np.array_equal(bits(m), bits(om)), b
        else:
            d = m.astype(np.float64) - om.astype(np.float64)
            assert float(np.sqrt(np.mean(d * d))) <= RMS_TOL
    e.close()
    eng.close()


@pytest.mark.parametrize("n_tracks", [24, 300])
@pytest.mark.parametrize("channels", [2, 1])
def test_one_launch_callback_device_formats(n_tracks, channels):
    """wbx_engine_process_interleaved: the conversion as the epilogue of the spread sum (one group: of the mix workgroup's
    sum kernel stand-in) — bytes equal to the batch render's sum-kernel epilogue, every format; 1024-frame blocks for mono"""
    K = 3
    block = 512 if channels == 2 else 1024
    spec = synth.make_session("cbfmt", n_tracks, n_blocks=K, amp=0.3 / np.sqrt(n_tracks / 24), seed=0xCBF, src_rate=44100, block=block)
    spec.channels = channels
    eng = build_engine(spec, max_blocks=1)
    ref = build_engine(spec, max_blocks=K, group_size=callback_group(n_tracks))
    for fmt in ("i16", "i24", "i24_x8", "i32", "f32"):
        ref.ctx.set_master_format(fmt)
        ref.play()
        ref.render(K)
        want = ref.ctx.fetch_interleaved(fmt).view(np.uint8).reshape(K, -1)
        ref.stop()
        eng.play()
        for b in range(K):
            got = eng.process_interleaved(fmt)
            if fmt == "i24":      # (the reference's packed writer leaves all but the first 3*F bytes of a block untouched)
                assert np.array_equal(got.view(np.uint8)[:3 * block], want[b][:3 * block]), (fmt, b)
            else:
                assert np.array_equal(got.view(np.uint8), want[b]), (fmt, b)
        eng.stop()
    eng.close()
    ref.close()


def test_one_launch_callback_parameter_and_clip_edits_between_blocks():
    """the device copy of the gains follows every parameter change; clip edits re-upload the lists; stop / play / tempo /
    playhead jumps go through the pinned patches — all between one-launch blocks"""
    spec = synth.make_session("cbedit", 130, src_rate=44100, n_blocks=40, seed=0xCBE)

    def vol(t, db):
        return lambda e, eng: (e.set_volume(t, db), eng.tracks[t].set_volume(db))

    def pan_mute(e, eng):
        e.set_pan(7, -0.4); eng.tracks[7].set_pan(-0.4)
        e.set_mute(100, True); eng.tracks[100].set_mute(True)

    def gain_and_clip(e, eng):
        e.set_clip_gain(3, 0, 0.5); eng.set_clip_gain(eng.tracks[3], 0, 0.5)
        e.add_audio_clip(5, 0.1, 0.3, 17.0, 9, 0.8, 0.7); eng.add_audio_clip(eng.tracks[5], "x", 0.1, 0.3, 17.0, 9, 0.8, 0.7)

    def stop_play(e, eng):
        e.stop(); eng.stop(); e.set_bpm(97.0); eng.set_bpm(97.0); e.set_playhead(0.05); eng.set_playhead_position(0.05); e.play(); eng.play()

    run_callback(spec, 14, edits={2: vol(0, -9.0), 3: vol(129, 2.0), 5: pan_mute, 7: gain_and_clip, 10: stop_play, 12: vol(64, -40.0)},
                 every_block_one_launch=False)
    assert sum(n.startswith("wbx::callback_kernel<") for n in return_names) >= 10, return_names


def test_one_launch_callback_repeats_blocks_that_queue_pre_render_rows():
    """clips shorter than a block put three stream calls into many blocks: the one-launch block reports a non-empty queue and
    the block is pre-rendered and mixed again through the three kernels (clips of 0.7 blocks: nearly every block; of 2.6
    blocks: some) — and a block without such a track-block is one launch again"""
    from test_gpu_parity import _boundary_session
    run_callback(_boundary_session(90, 8, 512, 0.7), 8, every_block_one_launch=False)
    assert any(n.startswith("wbx::mix_kernel<") for n in return_names), return_names
    run_callback(_boundary_session(90, 8, 512, 2.6), 8, every_block_one_launch=False)
    assert any(n.startswith("wbx::callback_kernel<") for n in return_names), return_names


def test_callback_grid_shrinks_from_more_than_the_device_to_a_spread_grid():
    """4608 tracks (288 workgroups: "the last one adds everything", only the FIRST ticket counter moves) — then tracks are
    deleted down to 4096 and 3000 (one workgroup per CU or fewer: the spread sum, whose second counter has a base of its own).
    With one base for both counters the first spread block after the large ones waited for a count that had wrapped: no
    report, a two-second time-out, WBX_ERR_DEVICE (round-4 advisor).  Every block against the oracle."""
    n0 = 4608
    spec = synth.make_session("cbshrink", n0, src_rate=44100, n_blocks=12, seed=0xC5B1)
    e = O.build_oracle_engine(spec)
    eng = build_engine(spec, max_blocks=1)
    out = W.AudioBuffer(spec.block, spec.channels)
    e.play()
    eng.play()
    n = n0
    spread_before = eng.callback_stats()[1]
    for b in range(10):
        if b in (3, 6):
            target = 4096 if b == 3 else 3000
            while n > target:          # from the back: the remaining tracks keep their slots
                n -= 1
                e.delete_track(n)
                eng.delete_track(n)
        om, _ = e.process()
        eng.process(None, out, float(spec.sample_rate))
        assert eng.ctx.kernel_name().startswith("wbx::callback_kernel<"), b
        m = np.stack(out.channel_buffers)
        d = m.astype(np.float64) - om.astype(np.float64)
        assert float(np.sqrt(np.mean(d * d))) <= RMS_TOL, b
        _, pk, _ = eng.ctx.fetch(peaks=True)
        assert np.array_equal(pk[0][:n], e.peaks()[:n, :spec.channels]), b
    launches, spread, give_ups, off = eng.callback_stats()
    assert launches == 10 and spread - spread_before == 7 and give_ups == 0 and off == 0
    e.close()
    eng.close()


@pytest.mark.parametrize("n_tracks", [300, 4096])
def test_callback_give_up_at_the_spread_barrier_mixes_the_block_again(monkeypatch, n_tracks):
    """WBX_CB_SPIN_BOUND=0: every workgroup that reaches the spread barrier before the last one gives up at once — what a grid
    that is not resident at once (a CU mask, a shared device) ends in after 50 ms.  The launch reports it, the block is mixed and
    summed again through three launches — same plan, same result — and the engine stops spreading."""
    monkeypatch.setenv("WBX_CB_SPIN_BOUND", "0")
    spec = synth.make_session("cbgiveup", n_tracks, src_rate=44100, n_blocks=6, seed=0xC5B2)
    e = O.build_oracle_engine(spec)
    e.enable_seglog()
    eng = build_engine(spec, max_blocks=1)
    out = W.AudioBuffer(spec.block, spec.channels)
    e.play()
    eng.play()
    for b in range(5):
        om, _ = e.process()
        eng.process(None, out, float(spec.sample_rate))
        m = np.stack(out.channel_buffers)
        d = m.astype(np.float64) - om.astype(np.float64)
        assert float(np.sqrt(np.mean(d * d))) <= RMS_TOL, b
        _, pk, _ = eng.ctx.fetch(peaks=True)
        assert np.array_equal(pk[0], e.peaks()[:, :spec.channels]), b
        if n_tracks <= 1025:
            assert plan_rows(eng.fetch_plan()) == oracle_rows(e, 0), b
    launches, spread, give_ups, off = eng.callback_stats()
    assert launches == 5 and spread == 1 and give_ups == 1 and off == 1     # block 0 gave up; blocks 1-4: the last workgroup adds
    e.close()
    eng.close()


def _dense_boundary_session(n_tracks, n_blocks, block=512):
    """fp32 tracks cut into clips of 0.23 blocks with gaps: four or five stream calls in most track-blocks — records in the
    sequencer's overflow pool and in the pre-render queue from block 0 on (the plan counters [0], [2], [3] are all in use)"""
    beat_frames = 48000 * 60.0 / 120.0
    total = (n_blocks + 1) * block
    samples, clips, vols, pans = [], [], [], []
    for t in range(n_tracks):
        samples.append(synth.SampleSpec(seed_track=t, channels=2, rate=[48000, 44100][t % 2], frames=int(total * 1.2) + 400, fmt="f32", amp=0.02))
        v, p = synth.track_params(0xD0B, t)
        vols.append(float(v))
        pans.append(float(p))
        L = 0.23 * block
        pos, k = -((t * 37) % 101) / 101.0 * L, 0
        while pos < total:
            a, b = max(pos, 0.0), pos + L * (0.6 if (t + k) % 3 == 0 else 1.0)
            if b > a:
                clips.append(synth.ClipSpec(t, a / beat_frames, b / beat_frames, start_offset=a * 0.9, gain=[1.0, 0.5, 1.7][k % 3]))
            pos += L
            k += 1
    return synth.SessionSpec(name="dense", n_tracks=n_tracks, seed=0xD0B, samples=samples, clips=clips, volumes_db=vols, pans=pans,
                             mutes=[False] * n_tracks, block=block)


@pytest.mark.parametrize("n_tracks", [300, 900])
def test_callback_give_up_with_the_plan_counters_in_use(monkeypatch, n_tracks):
    """(advisor, round 5) The give-up path on a block whose sequencer lanes allocate overflow-pool chunks and queue pre-render
    records: workgroup 0 of the spread launch gives up at once (WBX_CB_SPIN_BOUND=0) while other workgroups are still planning —
    it must neither publish nor clear the plan's counters (two tracks would share a pool chunk, status bits would be lost); the
    launch's reporter hands them to the host, un-cleared, and the block comes out right: stream calls, peaks and master equal
    to the oracle's in the give-up block and in every block after it."""
    monkeypatch.setenv("WBX_CB_SPIN_BOUND", "0")
    spec = _dense_boundary_session(n_tracks, 5)
    e = O.build_oracle_engine(spec)
    e.enable_seglog()
    eng = build_engine(spec, max_blocks=1)
    out = W.AudioBuffer(spec.block, spec.channels)
    e.play()
    eng.play()
    for b in range(4):
        om, _ = e.process()
        eng.process(None, out, float(spec.sample_rate))
        m = np.stack(out.channel_buffers)
        d = m.astype(np.float64) - om.astype(np.float64)
        assert float(np.sqrt(np.mean(d * d))) <= RMS_TOL, b
        _, pk, _ = eng.ctx.fetch(peaks=True)
        assert np.array_equal(pk[0], e.peaks()[:, :spec.channels]), b
        rows = oracle_rows(e, 0)
        assert max(sum(1 for r in rows if r[1] == t) for t in range(0, n_tracks, 7)) >= 3      # (the session does what it says)
        assert plan_rows(eng.fetch_plan()) == rows, b
    launches, spread, give_ups, off = eng.callback_stats()
    assert give_ups == 1 and off == 1, (launches, spread, give_ups, off)
    e.close()
    eng.close()


def test_callback_give_up_does_not_lose_a_plan_overflow(monkeypatch):
    """(advisor, round 5) ... and a status bit raised in the give-up block reaches the caller: 1100 tracks that each need an
    overflow-pool chunk against the 1024 chunks a max_blocks = 1 engine has — the block that gave up at the spread barrier and
    was mixed again reports WBX_ERR_OVERFLOW (-8) like any other (the redo's sum used to drop the already-cleared device
    counters over the host's copy: the overflow of exactly that block was never seen)."""
    monkeypatch.setenv("WBX_CB_SPIN_BOUND", "0")
    spec = _dense_boundary_session(1100, 3)
    eng = build_engine(spec, max_blocks=1)
    out = W.AudioBuffer(spec.block, spec.channels)
    eng.play()
    with pytest.raises(W.WbxError) as ei:
        eng.process(None, out, float(spec.sample_rate))
    assert ei.value.status == -8 and "overflow" in str(ei.value), str(ei.value)
    assert eng.callback_stats()[2] == 1           # (it was the give-up block)
    eng.close()


def test_xcd_layout_probe_and_its_fallback(monkeypatch):
    """wbx_create probes the workgroup-id -> XCD layout the chained pieces and the segmented sequencer rest on (MI355X: 8 XCDs,
    round-robin); a device where it does not hold (WBX_XCD_PROBE_FAIL=1 plays one) walks whole member lists and plans by one lane
    per track from its first render — same results, no WBX_ERR_DEVICE on the way"""
    spec = synth.make_session("xcd", 300, src_rate=44100, n_blocks=64, seed=0xC5B3)
    monkeypatch.setenv("WBX_EXACT_MIN_BLOCKS", "32")
    masters = {}
    for fail in ("0", "1"):
        monkeypatch.setenv("WBX_XCD_PROBE_FAIL", fail)
        eng = build_engine(spec, max_blocks=64)
        assert eng.ctx.xcd_count() == (8 if fail == "0" else 0)
        eng.play()
        eng.render(64)
        m, _, _ = eng.ctx.fetch()
        order = eng.ctx.render_order(64)
        assert order[2] is True and (order[0] > 1) == (fail == "0"), order     # the reference's order either way; chained only with the layout
        masters[fail] = m.copy()
        eng.close()
    assert np.array_equal(bits(masters["0"]), bits(masters["1"]))


def test_perf_measurer_follows_the_callback():
    """Engine::perf_measurer (engine.cpp:1577,1653; core/timing.h:54-67): every wbx_engine_process call feeds its own wall time
    and the block's period (engine.cpp:52).  The durations are the box's, the arithmetic is the reference's: the running figure
    the engine reports equals the oracle's PerformanceMeasurer statements replayed over the durations it reports, bit for bit;
    batch renders do not feed it."""
    spec = synth.make_session("perf", 64, src_rate=44100, n_blocks=40, seed=0xC5B4)
    eng = build_engine(spec, max_blocks=8)
    out = W.AudioBuffer(spec.block, spec.channels)
    L = O.lib()
    period = L.wbo_buffer_duration_ms(spec.block, spec.sample_rate)
    assert O.f64_bits(period) == O.f64_bits(W.lib().wbx_calc_buffer_period_ms(spec.block, spec.sample_rate)) and abs(period - 10.6667) < 1e-3
    assert eng.perf_usage() == (0.0, 0.0)
    eng.play()
    cur = 0.0
    for b in range(24):
        eng.process(None, out, float(spec.sample_rate))
        usage, last_ms = eng.perf_usage()
        assert 0.0 < last_ms < 5000.0, (b, last_ms)
        cur = L.wbo_perf_update(cur, last_ms, period)
        assert O.f64_bits(usage) == O.f64_bits(L.wbo_perf_get_usage(cur)), (b, usage, cur)
        if b == 11:
            eng.render(8)                    # (a batch render in between: not a callback, no update)
            eng.ctx.fetch()
            assert eng.perf_usage() == (usage, last_ms)
    assert 0.0 < usage < 1.0                 # a 64-track block takes a few dozen microseconds of its 10.7 ms
    eng.close()
