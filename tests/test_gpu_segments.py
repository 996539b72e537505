"""The sequencer cut along the time axis on the device (plan_seg_kernel: wbx_seq.h plan_segment /
plan_check_seams / plan_redo_track; tests/test_host_sim.py runs the same source on the CPU): long renders of sessions cut into clips are planned
by one lane per (track, segment) — results must be the one-walk plan's, i.e. the oracle's, bit for bit: stream-call log,
transport, peaks, master, and the state the NEXT render starts from."""
import os

import numpy as np
import pytest

import fuzz_util as FZ
import oracle_ffi as O
from whitebox_amd import synth
from whitebox_amd.engine import build_engine
from test_gpu_parity import _boundary_session, bits, check_against_oracle, plan_rows, run_oracle

pytestmark = pytest.mark.gpu


# WBX_SEGFUZZ_TO widens the seed ranges for a soak run (defaults: seeds 0..39 / 0..11)
@pytest.mark.parametrize("seg", ["1", "3", "8"])
@pytest.mark.parametrize("seed", range(0, int(os.environ.get("WBX_SEGFUZZ_TO", "40"))))
def test_random_sessions_planned_by_segments(monkeypatch, seed, seg):
    """the first fuzz generator's sessions (overlapping adds, sub-block clips, mid-session playheads, buses) with forced
    segments of 1 / 3 / 8 blocks: whatever the seam guesses do, everything equals the oracle"""
    monkeypatch.setenv("WBX_PLAN_SEG", seg)
    spec, n_blocks = FZ.random_session(seed)
    check_against_oracle(spec, n_blocks, expect_exact=True)


@pytest.mark.parametrize("kind", ["masked", "integer", "lean16", "everything"])
@pytest.mark.parametrize("seed", range(0, max(12, int(os.environ.get("WBX_SEGFUZZ_TO", "40")) // 3)))
def test_masked_row_sessions_planned_by_segments(monkeypatch, seed, kind):
    monkeypatch.setenv("WBX_PLAN_SEG", "2")
    spec, n_blocks = FZ.random_masked_session(seed, integer_unity=kind == "integer", lean16=kind == "lean16", everything=kind == "everything")
    check_against_oracle(spec, n_blocks, group_size=max(spec.n_tracks, 1) if spec.n_tracks <= 128 else 0, expect_exact=spec.n_tracks <= 128)


@pytest.mark.parametrize("block,clip_blocks,n_tracks,n_blocks", [(512, 5.3, 96, 160), (128, 5.3, 70, 256), (256, 1.3, 64, 130), (512, 20.0, 40, 192)])
def test_default_segments_on_long_cut_sessions(block, clip_blocks, n_tracks, n_blocks):
    """what the library does by itself: renders of >= 128 blocks of a session with tracks cut into clips are planned by segments
    (wbx_engine_sequencer_stats says so), no seam misses on back-to-back clips, and the second render — entered with the state
    the first one's last segments left — is right as well"""
    spec = _boundary_session(n_tracks, 2 * n_blocks, block, clip_blocks)
    om, opk, _, orows, otr = run_oracle(spec, 2 * n_blocks)
    eng = build_engine(spec, max_blocks=n_blocks, group_size=n_tracks)
    eng.play()
    for half in range(2):
        eng.render(n_blocks)
        m, pk, _ = eng.ctx.fetch(peaks=True)
        lo = half * n_blocks
        assert np.array_equal(bits(m), bits(om[lo:lo + n_blocks])), half
        assert np.array_equal(pk, opk[lo:lo + n_blocks, :, :spec.channels]), half
        got = [(b + lo,) + r[1:] for (b, *rest) in [tuple(x) for x in plan_rows(eng.fetch_plan())] for r in [(b, *rest)]]
        assert got == [r for r in orows if lo <= r[0] < lo + n_blocks], half
    renders, tracks_redone, segs_redone, segs = eng.sequencer_stats()
    assert renders == 2 and segs >= 2, (renders, segs)
    assert tracks_redone <= n_tracks // 8, (tracks_redone, segs_redone)   # (gaps and stretch changes: a few seams may miss)
    ph, sp, _ = eng.transport()
    assert (O.f64_bits(ph), O.f64_bits(sp)) == (O.f64_bits(otr[0]), O.f64_bits(otr[1]))
    eng.close()


def test_segments_off_while_a_clip_flag_is_set():
    """an edit that sets Clip::internal_state_changed (move_clip, engine.cpp:346-363) switches the renders back to one walk per
    track until the sequencer has passed — and cleared — the flag (it counts the set flags of the table down in host memory:
    lanes of one track would race for a flag), then they are segmented again; all of them equal the oracle"""
    n_tracks, K = 24, 128
    spec = _boundary_session(n_tracks, 4 * K, 512, 7.3, gaps=False)
    e = O.build_oracle_engine(spec)
    eng = build_engine(spec, max_blocks=K, group_size=n_tracks)
    e.play()
    eng.play()

    def step(tag):
        eng.render(K)
        m, pk, _ = eng.ctx.fetch(peaks=True)
        for b in range(K):
            om, _ = e.process()
            assert np.array_equal(bits(m[b]), bits(om)), (tag, b)
            assert np.array_equal(pk[b], e.peaks()[:, :2]), (tag, b)

    step("first")
    assert eng.sequencer_stats()[0] == 1
    ph = eng.transport()[0]
    cl = eng.clips(eng.tracks[3])
    nxt = [i for i, ci in enumerate(cl) if ci[0] > ph]
    assert len(nxt) >= 2
    i = nxt[0]                                      # the clip behind the playing one: reached during the next render
    delta = 0.2 * (cl[i + 1][0] - cl[i][1]) if cl[i + 1][0] > cl[i][1] else 0.0
    if delta == 0.0:                                # back-to-back clips: move it back onto the tail of the playing one instead
        delta = -0.1 * (cl[i][1] - cl[i][0])
    e.move_clip(3, i, delta)
    eng.move_clip(eng.tracks[3], i, delta)
    assert e.clips(3) == eng.clips(eng.tracks[3])
    step("after the edit")
    assert eng.sequencer_stats()[0] == 1          # one walk per track: a flag was set
    step("third")
    assert eng.sequencer_stats()[0] == 2          # ... and has been cleared: segments again
    step("fourth")
    assert eng.sequencer_stats()[0] == 3
    e.close()
    eng.close()
