"""Blocks between the shapes the mix instances are cut for (round 5).

The reference's settings dialog offers 32 … 4096 frames in powers of two (ui/settings.cpp:22-24), but the block the engine is
really driven with is the device's period: `g_audio_buffer_size = period_to_buffer_size(min_period)`, realigned to the back
end's buffer alignment (config.cpp:146-149,217-222; 32 frames for PulseAudio and at most 32 for WASAPI,
audio_io_pulseaudio.cpp:244-245, audio_io_wasapi.cpp:428).  WASAPI's shared mode grants 10 ms: 480 frames at 48 kHz, 441 ->
416 at 44.1 kHz, 960 / 1920 at 96 / 192 kHz.  Such a block takes the instance of the next shape above it — its surplus lanes
clone the block's last four frames (wbx_mix.h, MixArgs::lane_span) — instead of the general instance of earlier rounds
(0.29-0.33 of the roofline; now 0.57-0.63, tools/exp_blocks.sh).  Everything here against the oracle: stream calls, peaks, the
master bit for bit in the reference's order."""
import numpy as np
import pytest

import oracle_ffi as O
import whitebox_amd as W
from whitebox_amd import synth
from whitebox_amd.engine import build_engine
import test_gpu_parity as P

pytestmark = pytest.mark.gpu

PERIODS = [480, 416, 448, 960, 320, 1440, 224, 96, 36, 1920, 2400]     # multiples of 4 frames; 36: nothing like any shape


@pytest.mark.parametrize("block", PERIODS)
@pytest.mark.parametrize("channels", [2, 1])
def test_device_periods_resampled_session(block, channels):
    """c3's kind of session (44.1 kHz fp32 clips, gain + pan, a clip start and stop inside blocks) at a device period: the
    grouped order, and one group for all tracks (the reference's order: bit-exact)"""
    spec = synth.make_session("per", 150, seek=block >= 128, n_blocks=5, block=block, seed=0x9E0 + block, src_rate=44100)
    spec.channels = channels
    P.check_against_oracle(spec, 5, group_size=32)
    P.check_against_oracle(spec, 5, group_size=150, expect_exact=True)


@pytest.mark.parametrize("block,instance", [(480, "wbx::mix_kernel<2, true, 4, 0, 1, 1, 1, 256>"), (960, "wbx::mix_kernel<2, true, 4, 0, 1, 1, 1, 256>"),
                                            (224, "wbx::mix_kernel<2, true, 4, 0, 2, 1, 1, 256>"), (96, "wbx::mix_kernel<2, true, 4, 0, 4, 2, 1, 256>")])
def test_device_periods_take_the_instances_of_the_next_shape(block, instance):
    spec = synth.make_session("pern", 64, n_blocks=8, block=block, seed=0x9E1, src_rate=44100)
    eng = build_engine(spec, max_blocks=8)
    eng.play()
    eng.render(8)
    eng.ctx.fetch()
    assert eng.ctx.kernel_name() == instance
    eng.close()


@pytest.mark.parametrize("block,channels,clip_blocks", [(480, 2, 1.3), (480, 2, 0.7), (416, 2, 2.2), (960, 2, 1.3), (224, 2, 1.9), (96, 2, 1.3),
                                                        (480, 1, 1.3), (200, 1, 2.1)])
@pytest.mark.parametrize("masked", ["1", "0"])
def test_device_periods_clip_boundaries_in_the_hot_loop(monkeypatch, masked, block, channels, clip_blocks):
    """sessions cut into clips: the masked rows of the instances these blocks now take (a partial stream call ends at F, not at
    the end of the instance's lane space; the clone lanes lie inside or outside a call with the frames they clone)"""
    monkeypatch.setenv("WBX_MASKED_ROWS", masked)
    n_blocks = 9
    spec = P._boundary_session(40, n_blocks, block, clip_blocks, channels)
    P.check_against_oracle(spec, n_blocks, group_size=40, expect_exact=True)
    P.check_against_oracle(spec, n_blocks, group_size=16)


@pytest.mark.parametrize("block", [480, 416, 960])
@pytest.mark.parametrize("fmts,rates", [(("i16",), (44100, 48000)), (("i24",), (44100, 48000)), (("i16", "i24", "f32"), (44100, 96000, 48000)),
                                        (("i16",), (48000,))])
def test_device_periods_every_family(block, fmts, rates):
    """the 16-bit family, the everything family with and without per-frame taps, integer PCM at the session rate — cut into clips"""
    n_blocks = 9
    spec = P._boundary_session(40, n_blocks, block, 1.3)
    for i, smp in enumerate(spec.samples):
        smp.fmt = fmts[i % len(fmts)]
        smp.rate = rates[(i // 2) % len(rates)]
        smp.amp = 0.02 if smp.fmt == "f32" else 1.0
        smp.frames = int(smp.frames * 2.2)
    if rates == (48000,):
        for c in spec.clips:
            c.speed = 1.0
    spec.volumes_db = [v - (0.0 if spec.samples[2 * t].fmt == "f32" else 30.0) for t, v in enumerate(spec.volumes_db)]
    P.check_against_oracle(spec, n_blocks, group_size=40, expect_exact=True)


@pytest.mark.parametrize("block,n_buses", [(480, 6), (960, 3), (416, 0)])
def test_device_periods_sub_buses_and_many_groups(block, n_buses):
    spec = synth.make_session("perb", 384, seek=True, n_blocks=9, block=block, seed=0x9E2, n_buses=n_buses, src_rate=44100)
    P.check_against_oracle(spec, 9)


@pytest.mark.parametrize("block,channels,n_tracks", [(480, 2, 48), (480, 2, 300), (480, 2, 4096), (960, 1, 64), (416, 2, 17)])
def test_callback_at_a_device_period_is_one_launch(block, channels, n_tracks):
    """Engine::process at 480 frames — the period WASAPI's shared mode really grants — is the one-launch callback now (it took the
    512-frame shape's three launches before); 960 mono likewise; 416 stereo, too"""
    K = 6
    spec = synth.make_session("percb", n_tracks, seek=n_tracks <= 300, n_blocks=K, block=block, seed=0x9E3 + n_tracks, src_rate=44100)
    spec.channels = channels
    e = O.build_oracle_engine(spec)
    e.enable_seglog()
    eng = build_engine(spec, max_blocks=1)
    out = W.AudioBuffer(block, channels)
    e.play()
    eng.play()
    names = []
    for b in range(K):
        om, _ = e.process()
        eng.process(None, out, float(spec.sample_rate))
        names.append(eng.ctx.kernel_name())
        m = np.stack(out.channel_buffers)
        if n_tracks <= 64:
            assert np.array_equal(P.bits(m), P.bits(om)), b
        else:
            assert P.rms(m, om) <= P.RMS_TOL, b
        _, pk, _ = eng.ctx.fetch(peaks=True)
        assert np.array_equal(pk[0], e.peaks()[:, :channels]), b
        if n_tracks <= 300:
            assert P.plan_rows(eng.fetch_plan()) == P.oracle_rows(e, 0), b
    assert sum(n.startswith("wbx::callback_kernel<") for n in names) >= K - 2, names     # (a block with three stream calls is repeated through three launches)
    ph, sp, _ = eng.transport()
    assert (O.f64_bits(ph), O.f64_bits(sp)) == (O.f64_bits(e.playhead), O.f64_bits(e.sample_position))
    e.close()
    eng.close()


def test_long_chained_render_at_a_device_period():
    """1024 blocks of 480 frames, 1024 tracks: the chained order (running sums handed on between the 128-track pieces — only the
    lanes that own frames load and store them) against the whole-list walk of the same engine configuration, and the head
    against the oracle"""
    K, N = 1024, 1024
    spec = synth.make_session("perlong", N, n_blocks=K, block=480, seed=0x9E4, src_rate=44100)
    eng = build_engine(spec, max_blocks=K, device_synth=True)
    eng.play()
    eng.render(K)
    m, pk, _ = eng.ctx.fetch(peaks=True)
    order = eng.ctx.render_order(K)
    eng.close()
    om, opk, _, _, _ = P.run_oracle(spec, 4)
    assert np.array_equal(P.bits(m[:4]), P.bits(om)) and np.array_equal(pk[:4], opk[..., :2]), order
    eng2 = build_engine(spec, max_blocks=K, group_size=N, device_synth=True)
    eng2.play()
    eng2.render(K)
    m2, pk2, _ = eng2.ctx.fetch(peaks=True)
    eng2.close()
    assert np.array_equal(P.bits(m), P.bits(m2)) and np.array_equal(pk, pk2)


def test_general_instance_is_still_there(monkeypatch):
    """WBX_RAGGED=0: the general instance of earlier rounds (lane predicates, records per lane) — same results"""
    monkeypatch.setenv("WBX_RAGGED", "0")
    spec = synth.make_session("pergen", 96, seek=True, n_blocks=5, block=480, seed=0x9E5, src_rate=44100)
    P.check_against_oracle(spec, 5, group_size=96, expect_exact=True)
    eng = build_engine(spec, max_blocks=5)
    eng.play()
    eng.render(5)
    eng.ctx.fetch()
    assert eng.ctx.kernel_name() == "wbx::mix_kernel<2, false, 1, 1, 1, 1, 1, 256>"
    eng.close()
