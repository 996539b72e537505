"""The TAIL of long chained renders against the oracle (reference order: engine.cpp:1600-1617 — one thread, track after
track).  A render of >= 1024 blocks adds in that order as chained 128-track pieces: chain words, epoch tags, the XCD-level
hand-over and the bounded sum grid all act far behind the first blocks, which is where the head checks of earlier rounds
stopped.  tests/oracle_tail.py brings the oracle to the last block in seconds (sharded over the host's cores, clip audio
only where a compared block reads it) without changing one addition of the compared blocks."""
import numpy as np
import pytest

import oracle_tail as OT
from whitebox_amd import synth
from whitebox_amd.engine import build_engine, plan_rows_of_blocks

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def check_blocks(K):
    return sorted(set(list(range(127, K, 128)) + list(range(K - 8, K)) + [0, 1]))


def compare(eng, spec, K, check, renders=1):
    """render K blocks `renders` times in a row (the later ones start deep inside the session: epoch tags and plan buffers
    have been round the ring), compare the LAST render's check blocks with the oracle"""
    want = OT.oracle_at_blocks(OT.descs_from_spec(spec), [(renders - 1) * K + b for b in check], block=spec.block,
                               channels=spec.channels, sample_rate=spec.sample_rate, bpm=spec.bpm, n_buses=spec.n_buses)
    eng.play()
    for _ in range(renders):
        eng.render(K)
    m, pk, _ = eng.ctx.fetch(peaks=True)
    rows = plan_rows_of_blocks(eng.fetch_plan_array(), check)
    for b in check:
        om, opk, orows = want[(renders - 1) * K + b]
        assert np.array_equal(bits(m[b]), bits(om)), ("master", b, float(np.abs(m[b] - om).max()))
        assert np.array_equal(bits(pk[b]), bits(opk[:, :spec.channels])), ("peaks", b)
        assert rows[b] == orows, ("plan rows", b)


@pytest.mark.parametrize("name,kw,cut", [("c3", dict(src_rate=44100), 0.0), ("c4", dict(n_buses=64), 0.0),
                                         ("c3_cut5.3", dict(src_rate=44100), 5.3)])
def test_tail_of_a_1024_block_chained_render_is_the_oracles(name, kw, cut):
    """BASELINE configs[2] / configs[3] at full width and the headline session cut into 5.3-block clips, K = 1024 (the
    library's default for that length: chained pieces): blocks 1016-1023, one block per 128 and the head — master bit for
    bit, per-track peaks, the sequencer's stream calls"""
    N, K = 4096, 1024
    spec = synth.make_session(name, N, n_blocks=K, seed=0x5EED0003, **kw)
    if cut:
        spec = synth.cut_into_clips(spec, cut, K)
    eng = build_engine(spec, max_blocks=K, device_synth=True)
    compare(eng, spec, K, check_blocks(K))
    assert eng.ctx.render_order(K)[2] is True
    if not kw.get("n_buses"):
        assert eng.ctx.render_order(K)[:2] == (32, 128)          # 32 chained pieces of 128 tracks
    eng.close()


def test_tail_of_the_third_2048_block_render():
    """the bench's operating point (2048-block renders) three renders in: the epoch tags have advanced, the plan / partial
    rings have been round once, the transport sits 4096 blocks into the session"""
    N, K = 2048, 2048
    spec = synth.make_session("c3", N, src_rate=44100, n_blocks=3 * K, seed=0x5EED0003)
    eng = build_engine(spec, max_blocks=K, device_synth=True)
    compare(eng, spec, K, [0, 1, 1023, 1024, 2040, 2041, 2046, 2047], renders=3)
    eng.close()
