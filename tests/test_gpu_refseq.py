"""The PRODUCT against the reference's own recorded answers (tests/golden/sequencer.npz: what Track::process_event /
Track::process / Engine::process of oracle/_ref/wbref_engine — the reference's code, cut out of its sources where they lie and
compiled unmodified — answered to 40 session scripts; oracle/gen_golden.py sequencer).  Through the C ABI's Engine surface, as
the audio callback (one wbx_engine_process per block) and as batch renders: master bit for bit, playhead / sample_position bit
for bit, the running VU maxima, the clip lists after every edit.  Edits the reference's cut could not take (they need
reserve_track_region) are skipped here exactly where the recording says they were refused."""
import numpy as np
import pytest

import oracle_ffi as O
import whitebox_amd as W
from whitebox_amd.engine import AudioBuffer, Engine

pytestmark = pytest.mark.gpu


def _cases():
    from test_oracle_golden import sequencer_golden_cases
    return [c for c in sequencer_golden_cases() if c[1].block % 4 == 0]


def _replay(s, want, batch):
    max_run = max([o[1] for o in s.ops if o[0] == "run"] + [1])
    eng = Engine(8, s.block, s.rate, s.channels, max_blocks=max_run if batch else 1)
    out = AudioBuffer(s.block, s.channels)
    level = {}                      # Track object -> running maximum per channel (the reference's VUMeter::level is never read here)
    wi = 0
    blocks = 0
    rate_now = [s.rate]
    for o in s.ops:
        k = o[0]
        rec = want[wi]
        wi += 1
        if k == "run":
            assert rec[0] == "run" and len(rec[1]) == o[1]
            if batch:
                eng.render(o[1])
                m = eng.ctx.fetch()[0]          # [K][C][F]
            for b, br in enumerate(rec[1]):
                if batch:
                    got = np.ascontiguousarray(m[b]).view(np.uint32)
                else:
                    eng.process(None, out, float(rate_now[0]))
                    got = np.stack([out.get_write_pointer(c) for c in range(s.channels)]).view(np.uint32)
                    ph, sp, _pl = eng.transport()
                    assert (O.f64_bits(ph), O.f64_bits(sp)) == (br["playhead"], br["sample_position"]), (br["block"], ph, sp)
                    lv = eng.levels()
                    for t, tr in enumerate(eng.tracks):
                        cur = level.setdefault(tr, np.zeros(2, np.float32))
                        for c in range(s.channels):
                            cur[c] = max(cur[c], lv[t, c])
                        ref_lv = np.array(br["tracks"][t]["level"], np.uint32).view(np.float32)
                        assert np.array_equal(cur[:s.channels], ref_lv[:s.channels]), (br["block"], t, cur, ref_lv)
                assert np.array_equal(got, br["master"]), (br["block"], np.argwhere(got != br["master"])[:4].tolist())
                blocks += 1
            if batch:
                ph, sp, _pl = eng.transport()
                assert (O.f64_bits(ph), O.f64_bits(sp)) == (rec[1][-1]["playhead"], rec[1][-1]["sample_position"])
            continue
        if k == "query":
            continue                # (a question to the reference's Track::query_clip_by_range: the oracle answers it, the engine has no such call)
        if k == "clips":
            assert rec[0] == "clips" and len(rec[1]) == len(eng.tracks)
            for t, tr in enumerate(eng.tracks):
                mine = [(O.f64_bits(c[0]), O.f64_bits(c[1]), O.f64_bits(c[2]), O.f64_bits(c[3]), O.f32_bits(c[4]), c[5])
                        for c in eng.clips(tr)]
                assert mine == rec[1][t], (t, mine, rec[1][t])
            continue
        assert rec[0] == "op"
        if rec[1] != 1:
            continue                # refused by the reference's cut (needs reserve_track_region) or an index out of range
        if k in ("cfg",):
            pass
        elif k == "bpm":
            eng.set_bpm(o[1])
        elif k == "seek":
            eng.set_playhead_position(o[1])
        elif k == "rate":
            eng.set_audio_channel_config(0, s.channels, s.block, int(o[1]))
            rate_now[0] = int(o[1])
        elif k == "play":
            eng.play()
        elif k == "stop":
            eng.stop()
        elif k == "sample":
            fmt, ch, rate, frames, data = s.samples[o[1]][:5]
            eng.add_sample(fmt, rate, [np.ascontiguousarray(d[:frames]) for d in data], frames)
        elif k == "track":
            eng.add_track("t")
        elif k == "vol":
            eng.tracks[o[1]].set_volume(o[2])
        elif k == "pan":
            eng.tracks[o[1]].set_pan(o[2])
        elif k == "mute":
            eng.tracks[o[1]].set_mute(bool(o[2]))
        elif k == "clip":
            _, t, mn, mx, so, si, sp, g = o
            eng.add_audio_clip(eng.tracks[t], "c", mn, mx, so, si, sp, g)
        elif k == "delclip":
            eng.delete_clip(eng.tracks[o[1]], o[2])
        elif k == "gain":
            eng.set_clip_gain(eng.tracks[o[1]], o[2], o[3])
        elif k == "move":
            eng.move_clip(eng.tracks[o[1]], o[2], o[3])
        elif k == "deltrack":
            eng.delete_track(o[1])
        elif k == "movetrack":
            eng.move_track(o[1], o[2])
        elif k == "solo":
            eng.solo_track(o[1])
        else:
            raise AssertionError(k)
    eng.close()
    return blocks


@pytest.mark.parametrize("batch", [False, True], ids=["callback", "render"])
def test_product_equals_the_reference_recordings(batch):
    n = blocks = 0
    for name, s, want in _cases():
        try:
            blocks += _replay(s, want, batch)
        except AssertionError as e:
            raise AssertionError(f"{name} ({'render' if batch else 'callback'}): {e}") from e
        n += 1
    assert n >= 28 and blocks > 350, (n, blocks)


@pytest.mark.parametrize("kind", ["static", "controls", "edits", "dense", "wild"])
def test_product_equals_the_live_reference(kind):
    """the same comparison against the reference executable itself where it travelled with the tree (oracle/_ref/wbref_engine is a
    prebuilt test artefact like liboracle.so; nothing of /root/reference is read): fresh seeds, scripts of tests/seq_sessions.py
    run through the reference on the host and replayed on the device — callback for even seeds, batch renders for odd ones.
    WBX_REFSEQ_GPU_SEEDS widens the range (soak runs)."""
    import os
    import ref_engine as R
    import seq_sessions as S
    if not R.available():
        pytest.skip("oracle/_ref/wbref_engine did not travel (built only where /root/reference exists)")
    n_seeds = int(os.environ.get("WBX_REFSEQ_GPU_SEEDS", "12"))
    first = int(os.environ.get("WBX_REFSEQ_GPU_FROM", "5000"))
    done = blocks = 0
    for seed in range(first, first + n_seeds):
        s = S.session_script(seed, kind)
        if s.block % 4:
            continue                      # the product takes blocks of a multiple of 4 frames
        try:
            R.run_oracle(s)               # only to learn whether the script reaches the reference's event_length wrap
        except R.Wrapped:
            continue
        want = R.run_reference(s)
        try:
            blocks += _replay(s, want, batch=bool(seed & 1))
        except AssertionError as e:
            raise AssertionError(f"{kind} seed {seed} ({'render' if seed & 1 else 'callback'}): {e}") from e
        done += 1
    assert done >= n_seeds // 3, (done, n_seeds)


def _levels_running_max(pk):
    """per-track running maximum of the block peaks [K][N][C] -> what VUMeter::level holds when nobody read it ([N][2] bit patterns,
    a mono session's second meter stays 0)"""
    lv = np.zeros((pk.shape[1], 2), np.float32)
    lv[:, :pk.shape[2]] = pk.max(axis=0)
    return lv.view(np.uint32)


@pytest.mark.parametrize("which", ["c1", "c2", "c3", "c3seek", "c2seek", "c5"])
def test_baseline_configs_equal_the_reference_recordings(monkeypatch, which):
    """BASELINE.json's configurations on the device against what the REFERENCE'S OWN Engine::process rendered
    (tests/golden/baseline_ref.npz, oracle/gen_golden.py baseline): master bit for bit in the reference's summation order
    (one group; config 3: chained 128-track pieces; config 5: the chain of 8 engines that WBX_DIST_CHAIN runs across GPUs),
    playhead and sample_position bit for bit, every track's running VU maximum."""
    from test_oracle_golden import baseline_ref_cases
    from whitebox_amd.engine import build_engine
    for name, spec, K, rec in baseline_ref_cases():
        if name != which:
            continue
        n = spec.n_tracks
        if name == "c5":
            from test_dist_gloo import _shard_spec
            from whitebox_amd.dist import PinnedBuffer, shard_tracks
            world = 8
            running = PinnedBuffer(K * 2 * 512)
            for rank in range(world):
                first, count = shard_tracks(n, world, rank)
                eng = build_engine(_shard_spec(spec, first, count), max_blocks=K, group_size=count)
                eng.ctx.set_clamp(rank == world - 1)
                eng.ctx.set_master_target(running.ptr)
                eng.ctx.set_master_init(running.ptr if rank else None)
                eng.play()
                eng.render(K)
                _, pk, _ = eng.ctx.fetch(peaks=True)
                assert np.array_equal(_levels_running_max(pk), rec["level"][first:first + count]), rank
                ph, sp, _ = eng.transport()
                assert (O.f64_bits(ph), O.f64_bits(sp)) == tuple(int(x) for x in rec["transport"][K - 1])
                eng.close()
            got = running.array.reshape(K, 2, 512).copy()
            running.close()
            assert np.array_equal(got.view(np.uint32), rec["master"])
            return
        if n > 512:
            monkeypatch.setenv("WBX_EXACT_MIN_BLOCKS", str(K))     # the chained order, as renders of >= 1024 blocks take it
        eng = build_engine(spec, max_blocks=K, group_size=0 if n > 512 else n)
        eng.play()
        eng.render(K)
        m, pk, _ = eng.ctx.fetch(peaks=True)
        assert np.array_equal(np.ascontiguousarray(m).view(np.uint32), rec["master"]), name
        ph, sp, _ = eng.transport()
        assert (O.f64_bits(ph), O.f64_bits(sp)) == tuple(int(x) for x in rec["transport"][K - 1]), name
        assert np.array_equal(_levels_running_max(pk), rec["level"]), name
        eng.close()
        return
    raise AssertionError(which)


def test_config4_equals_the_reference_composition():
    """configuration 4 on the device against SURVEY A13's oracle formed from the reference's own functions (64 reference engines,
    one per bus, added with the reference's AudioBuffer::mix: tests/golden/baseline_ref.npz c4.*): the un-clamped master
    (wbx_set_clamp(0)) and all 64 bus sums bit for bit."""
    import os
    import zlib
    import golden_util as G
    from whitebox_amd import synth
    from whitebox_amd.engine import build_engine
    g = np.load(os.path.join(G.GOLDEN, "baseline_ref.npz"))
    spec = synth.make_session("c4", 4096, n_buses=64, n_blocks=4, seed=0x5EED0004)
    eng = build_engine(spec, max_blocks=4)
    eng.ctx.set_clamp(False)
    eng.play()
    eng.render(4)
    m, _, bus = eng.ctx.fetch(buses=True)
    assert np.array_equal(np.ascontiguousarray(m).view(np.uint32), g["c4.master_unclamped"])
    for b in range(4):
        assert [zlib.crc32(np.ascontiguousarray(bus[b, k]).view(np.uint32).tobytes()) for k in range(64)] == [int(x) for x in g["c4.bus_crc"][b]], b
    ph, sp, _ = eng.transport()
    assert (O.f64_bits(ph), O.f64_bits(sp)) == tuple(int(x) for x in g["c4.transport"][3])
    eng.close()
