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
FROM = int(os.environ.get("WBX_REFSEQ_FROM", "0"))       # soak runs: a seed range no earlier run has seen


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
    compared, wrapped, blocks = _differential(kind, range(FROM, FROM + N))
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
        ops = [o for o in s.ops if o[0] not in ("run", "clips", "query")]
        st = [r[1] for r in ref if r[0] == "op"]
        assert len(ops) == len(st)
        for o, x in zip(ops, st):
            if o[0] in ("clip", "move", "delclip", "gain"):
                taken += x == 1
                refused += x == 0
                bad += x == 2
    assert taken > 100 and refused > 10 and bad > 5, (taken, refused, bad)


CUTS = {   # build output under oracle/_ref/ -> the reference source it must be a verbatim, contiguous region of
    "eng/vu_meter_h_body.inc": "engine/vu_meter.h", "eng/track_h_body.inc": "engine/track.h",
    "eng/audio_record_h_body.inc": "engine/audio_record.h", "eng/engine_h_body.inc": "engine/engine.h",
    "eng/track_cpp_body.inc": "engine/track.cpp",
    "eng/engine_r1.inc": "engine/engine.cpp", "eng/engine_r2.inc": "engine/engine.cpp", "eng/engine_r3.inc": "engine/engine.cpp",
    "eng/engine_r4.inc": "engine/engine.cpp", "eng/engine_r5.inc": "engine/engine.cpp", "eng/engine_r6.inc": "engine/engine.cpp",
    "eng/assets_r1.inc": "engine/assets_table.cpp", "eng/assets_r2.inc": "engine/assets_table.cpp",
    "eng/assets_r3.inc": "engine/assets_table.cpp", "eng/sample_r1.inc": "dsp/sample.cpp",
    "deinterleave_impl.inc": "dsp/sample.cpp", "mip_impl.inc": "gfx/waveform_visual.cpp", "vu_meter_struct.inc": "engine/vu_meter.h",
}


def test_every_cut_is_a_verbatim_region_of_the_reference(exe):
    """what oracle/ref_*_driver.cpp compile is the reference's text and nothing else: every build output of the recipe is, byte
    for byte, ONE contiguous region of the source file it was cut from; regions of one file do not overlap; no `Log::` line
    survives in a cut except inside track.cpp's own `#if WB_DBG_LOG_*` blocks and audio_record.h's comment"""
    import re
    ref_root = "/root/reference/src"
    ranges = {}
    for inc, src in CUTS.items():
        text = open(os.path.join(R.O.ORACLE_DIR, "_ref", inc)).read()
        whole = open(os.path.join(ref_root, src)).read()
        at = whole.find(text)
        if at < 0 and whole.endswith(text[:-1]):     # (a region that runs to the end of a file without a final newline: awk adds one)
            at = len(whole) - len(text) + 1
        assert at >= 0 and text.strip(), (inc, "is not a verbatim region of", src)
        first = whole.count("\n", 0, at) + 1
        last = first + text.count("\n") - 1
        ranges.setdefault(src, []).append((first, last, inc))
        for m in re.finditer(r"^.*Log::.*$", text, re.M):
            line = m.group(0)
            if line.lstrip().startswith("//"):
                continue
            before = text[:m.start()]
            depth = len(re.findall(r"^#if", before, re.M)) - len(re.findall(r"^#endif", before, re.M))
            assert src == "engine/track.cpp" and depth > 0, (inc, line)
    for src, rs in ranges.items():
        if src in ("engine/vu_meter.h", "dsp/sample.cpp"):
            continue            # (two recipes cut these files for two drivers: the struct / the whole body; the transposition / the Sample members)
        rs.sort()
        for (a0, a1, _), (b0, b1, _) in zip(rs, rs[1:]):
            assert a1 < b0, (src, rs)
    print("\n".join(f"{src}:{a}-{b}  ({inc})" for src, rs in sorted(ranges.items()) for a, b, inc in sorted(rs)))
