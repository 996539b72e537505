"""The reference's two-thread contract on the real engine (SURVEY §8(b) "Threading"): a UI thread hammers
Track::set_volume / set_pan / set_mute (lock-free SPSC rings) and clip edits (editor lock) while the audio thread
renders block after block through Engine::process.  The library reports, per block, how many locked edits it had seen
and how many parameter messages it had drained per track; replaying exactly that message order into the oracle must
reproduce every block bit for bit (master, peaks, sequencer plan)."""
import threading

import numpy as np
import pytest

import oracle_ffi as O
import whitebox_amd as W
from whitebox_amd import synth
from whitebox_amd.engine import build_engine

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("src_rate,block", [(44100, 512), (48000, 128)])
def test_ui_thread_hammers_while_audio_thread_renders(src_rate, block):
    N, B = 24, 160
    spec = synth.make_session("threads", N, seek=True, src_rate=src_rate, n_blocks=B + 4, seed=0x7EAD, block=block)
    eng = build_engine(spec, max_blocks=1)      # group_size 64 >= N: the reference's summation order
    e = O.build_oracle_engine(spec)
    eng.play()
    e.play()
    out = W.AudioBuffer(spec.block, spec.channels)
    eng.process(None, out, float(spec.sample_rate))           # block 0 drains the set-up messages on both sides
    om, _ = e.process()
    assert np.array_equal(bits(np.stack(out.channel_buffers)), bits(om))
    seen0, drained0 = eng.thread_stats()

    param_log = [[] for _ in range(N)]       # per track, in push order: (kind, value)
    edit_log = []                             # in issue order: (name, args)
    stop = threading.Event()
    beat = 24000.0

    def ui():
        rng = np.random.default_rng(0x5157)
        i = 0
        while not stop.is_set():
            t = int(rng.integers(0, N))
            k = int(rng.integers(0, 10))
            if k < 4:
                v = float(np.float32(rng.uniform(-40, 3)))
                param_log[t].append(("v", v))
                eng.tracks[t].set_volume(v)
            elif k < 7:
                v = float(np.float32(rng.uniform(-1, 1)))
                param_log[t].append(("p", v))
                eng.tracks[t].set_pan(v)
            elif k < 8:
                v = bool(rng.integers(0, 2))
                param_log[t].append(("m", v))
                eng.tracks[t].set_mute(v)
            else:
                n = len(eng.clips(eng.tracks[t]))
                if n == 0:
                    continue
                ci = int(rng.integers(0, n))
                if k == 8:
                    g = float(np.float32(rng.uniform(0.2, 1.2)))
                    edit_log.append(("gain", (t, ci, g)))
                    eng.set_clip_gain(eng.tracks[t], ci, g)
                else:
                    rel = float(rng.normal(0, 700)) / beat
                    edit_log.append(("move", (t, ci, rel)))
                    eng.move_clip(eng.tracks[t], ci, rel)
            i += 1

    th = threading.Thread(target=ui)
    th.start()
    blocks = []
    try:
        for b in range(B):
            eng.process(None, out, float(spec.sample_rate))
            seen, drained = eng.thread_stats()
            _, pk, _ = eng.ctx.fetch(peaks=True)
            blocks.append((np.stack(out.channel_buffers).copy(), pk[0].copy(), seen, drained))
    finally:
        stop.set()
        th.join()
    assert sum(len(p) for p in param_log) > 50 and len(edit_log) > 5, "the UI thread did not get to run"

    # replay into the oracle: before block b, the edits and the per-track messages block b had taken
    prev_seen, prev_dr = seen0, drained0
    for b, (m, pk, seen, drained) in enumerate(blocks):
        assert seen >= prev_seen
        for name, args in edit_log[prev_seen - seen0:seen - seen0]:
            if name == "gain":
                assert e.set_clip_gain(*args) == 0
            else:
                assert e.move_clip(*args) == 0
        for t in range(N):
            for kind, v in param_log[t][prev_dr[t] - drained0[t]:drained[t] - drained0[t]]:
                {"v": e.set_volume, "p": e.set_pan, "m": e.set_mute}[kind](t, v)
        om, _ = e.process()
        assert np.array_equal(bits(m), bits(om)), (b, seen - seen0)
        assert np.array_equal(pk, e.peaks()[..., :spec.channels]), b
        prev_seen, prev_dr = seen, drained
    e.close()
    eng.close()


def test_parameter_ring_producer_waits_for_the_audio_thread():
    """63 messages fit a track's ring; the 64th push yields until the audio thread's next block drains it
    (core/queue.h:179-182) — the producer must come back once a block has run."""
    spec = synth.make_session("ringfull", 2, n_blocks=4, seed=0x7EAE)
    eng = build_engine(spec, max_blocks=1)
    eng.play()
    out = W.AudioBuffer(spec.block, spec.channels)
    eng.process(None, out, float(spec.sample_rate))
    done = threading.Event()

    def ui():
        for k in range(150):                      # more than two ring-fulls
            eng.tracks[0].set_volume(-float(k % 30))
        done.set()

    th = threading.Thread(target=ui)
    th.start()
    for _ in range(400):
        eng.process(None, out, float(spec.sample_rate))
        if done.is_set():
            break
    th.join(timeout=30)
    assert done.is_set()
    eng.process(None, out, float(spec.sample_rate))
    _, drained = eng.thread_stats()
    assert drained[0] == 5 + 150
    eng.close()
