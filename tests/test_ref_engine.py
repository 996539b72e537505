"""The oracle's clip sequencer and block driver against the REFERENCE'S OWN (oracle/_ref/wbref_engine: Track::process_event,
Track::process, Engine::process, add_audio_clip / add_to_cliplist / delete_clip / move_clip / set_clip_gain, delete_track /
move_track / solo_track ... cut out of engine/track.cpp and engine/engine.cpp where they lie and compiled unmodified — see
oracle/ref_engine_driver.cpp for what the cut holds and what it cannot).  Compared per block, bit for bit: the master, playhead,
sample_position, every track's AudioEvent list (type, buffer_offset, time, speed, sample_offset), current event type, sampler
speed / offset, VU levels; after edits the clip lists; and for every operation whether the reference took it.  This container
only (-m ref): tests/golden/sequencer.npz carries the reference's answers everywhere else."""
import os

import pytest

import ref_engine as R
import seq_sessions as S

pytestmark = pytest.mark.ref

N = int(os.environ.get("WBX_REFSEQ_SEEDS", "60"))


@pytest.fixture(scope="module")
def exe():
    if not R.available():
        pytest.skip("oracle/_ref/wbref_engine not built (no /root/reference here)")


def _differential(kind, seeds):
    wrapped = compared = blocks = 0
    for seed in seeds:
        s = S.session_script(seed, kind)
        try:
            orc = R.run_oracle(s)
        except R.Wrapped:
            wrapped += 1          # the reference would write past its block buffer (track.cpp:669): nothing to compare
            continue
        ref = R.run_reference(s)
        d = R.compare(ref, orc, f"{kind} seed {seed}")
        assert d is None, d
        compared += 1
        blocks += sum(len(r[1]) for r in ref if r[0] == "run")
    return compared, wrapped, blocks


@pytest.mark.parametrize("kind", ["static", "controls", "edits", "dense", "wild"])
def test_oracle_sequencer_equals_the_reference(exe, kind):
    compared, wrapped, blocks = _differential(kind, range(N))
    assert compared >= N * (0.5 if kind in ("dense", "wild") else 0.8), (compared, wrapped)
    print(f"{kind}: {compared} sessions / {blocks} blocks equal, {wrapped} left out (event_length wrap)")


def test_refusals_and_takes_are_both_exercised(exe):
    """the edit scripts reach both sides of the driver's gate: adds / moves the reference took and ones that would have
    needed reserve_track_region (refused on both sides alike)"""
    taken = refused = bad = 0
    for seed in range(30):
        s = S.session_script(seed, "edits")
        try:
            R.run_oracle(s)
        except R.Wrapped:
            continue
        ref = R.run_reference(s)
        ops = [o for o in s.ops if o[0] != "run" and o[0] != "clips"]
        st = [r[1] for r in ref if r[0] == "op"]
        assert len(ops) == len(st)
        for o, x in zip(ops, st):
            if o[0] in ("clip", "move", "delclip", "gain"):
                taken += x == 1
                refused += x == 0
                bad += x == 2
    assert taken > 100 and refused > 10 and bad > 5, (taken, refused, bad)
