"""Seeded session scripts for the differential against the reference's own sequencer / block driver (tests/ref_engine.py).
Shared by the `-m ref` test (oracle against oracle/_ref/wbref_engine, this container), by oracle/gen_golden.py (which records
the reference's answers into tests/golden/sequencer.npz) and by the tests that replay the recorded answers anywhere.

What a script may NOT do is what the compiled reference has no defined behaviour for: a mono clip resampled into a
stereo session (Q1: the linear path indexes the channel array without `% channels`), negative speeds or start offsets (Q12),
and — found per session, not by construction — the event_length wrap of track.cpp:669 (ref_engine.Wrapped)."""
import numpy as np

from whitebox_amd import synth

import ref_engine as R

PLAIN_RATES = [22050, 44100, 48000, 96000]
PLAIN_SPEEDS = [1.0, 1.0, 1.0, 0.5, 0.8, 0.91875, 0.999, 1.0625, 1.9, 0.3, 4.0]
WILD_RATES = [8000, 11025, 22050, 44100, 96000, 192000]
WILD_SPEEDS = [1.0, 0.01, 0.1, 0.9999999, 1.0000001, 3.99, 8.0, 33.0]
RATES, SPEEDS = PLAIN_RATES, PLAIN_SPEEDS


def _samples(rng, s: R.Script, seed, n, session_rate, out_channels, stereo_only=False):
    """n samples of every storage format; returns per sample whether it may be resampled (Q1)"""
    free = []
    for i in range(n):
        fmt = str(rng.choice(["f32", "f32", "i16", "i24", "i32"]))
        ch = int(rng.integers(1, 3))
        if stereo_only and out_channels == 2:
            ch = 2                         # the session rate may change in mid-play: a mono clip would end up resampled (Q1)
        rate = int(rng.choice(RATES + [session_rate] * 3))
        if ch == 1 and out_channels == 2:
            rate = session_rate            # a mono clip in a stereo session: unity path only
        frames = int(rng.choice([int(rng.integers(5, 300)), int(rng.integers(300, 6000)), int(rng.integers(6000, 30000))]))
        spec = synth.SessionSpec(name="s", n_tracks=1, seed=seed, samples=[synth.SampleSpec(i, ch, rate, frames, fmt,
                                                                                          0.2 if fmt == "f32" else 1.0)],
                                 clips=[], volumes_db=[0.0], pans=[0.0], mutes=[False])
        s.add_sample(fmt, ch, rate, frames, spec.sample_data(0), gen=(seed, i, 0.2 if fmt == "f32" else 1.0))
        free.append(not (ch == 1 and out_channels == 2))
    return free


def _clip_args(rng, t, si, resample_ok, pos, length, frames):
    speed = float(rng.choice(SPEEDS)) if resample_ok else 1.0
    so = float(rng.choice([0, 0, int(rng.integers(0, 400)), int(rng.integers(0, frames + 50))]))
    gain = float(np.float32(rng.choice([1.0, 1.0, 0.5, 1.3, 0.0])))
    return ("clip", t, float(pos), float(pos + length), so, si, speed, gain)


def session_script(seed, kind):
    """kind: 'static' (clip layouts only), 'controls' (transport / parameter / track operations between blocks), 'edits' (clip
    adds into free space, deletes aimed at the sounding clip, gains, moves), 'dense' (back-to-back clips, edges on block edges)"""
    rng = np.random.default_rng([seed, {"static": 1, "controls": 2, "edits": 3, "dense": 4, "wild": 5}[kind]])
    wild = kind == "wild"           # the corners of what the API accepts, all at once; otherwise an 'edits' script
    out_ch = int(rng.choice([1, 2, 2, 2]))
    block = int(rng.choice([32, 100, 128, 440, 1000, 2048] if wild else [64, 96, 128, 200, 256, 333, 512, 1024]))
    rate = int(rng.choice([22050, 32000, 88200, 192000, 48000] if wild else [44100, 48000, 48000, 96000]))
    bpm = float(rng.choice([20.0, 333.3, 999.0, 120.0] if wild else [120.0, 97.0, 140.5, 61.3, 174.0]))
    global RATES, SPEEDS
    RATES, SPEEDS = (WILD_RATES, WILD_SPEEDS) if wild else (PLAIN_RATES, PLAIN_SPEEDS)
    s = R.Script(out_ch, block, rate, bpm)
    n_tracks = int(rng.integers(1, 7))
    n_blocks = int(rng.integers(6, 25))
    beat_frames = rate * 60.0 / bpm
    total = n_blocks * block / beat_frames          # beats the session plays
    unit = block / beat_frames                      # beats per block
    resample_ok = _samples(rng, s, 0x5E90000 + seed, n_tracks, rate, out_ch, stereo_only=kind == "controls")
    for t in range(n_tracks):
        s.op("track")
        s.op("vol", t, float(np.float32(rng.uniform(-30, 3))))
        s.op("pan", t, float(np.float32(rng.uniform(-1, 1))))
        if rng.random() < 0.1:
            s.op("mute", t, 1)
    start = float(rng.choice([0.0, 0.0, total * 0.15]))
    # clip layouts: sequential per track (touching or with gaps), so that an add never needs reserve_track_region
    for t in range(n_tracks):
        si = int(rng.integers(0, n_tracks))
        frames = s.samples[si][3]
        pos = -0.2 * total * rng.random() if rng.random() < 0.3 else total * rng.random() * 0.3
        n_clips = int(rng.integers(0, 5)) if kind != "dense" else int(rng.integers(5, 40))
        for _ in range(n_clips):
            if wild and rng.random() < 0.4:           # sub-frame and few-frame clips
                length = float(rng.choice([0.3, 1.0, 2.5, 17.0])) / beat_frames
                s.op(*_clip_args(rng, t, si, resample_ok[si], pos, length, frames))
                pos += length + (0.0 if rng.random() < 0.5 else unit * rng.random())
                continue
            if kind == "dense":
                length = unit * float(rng.choice([0.2, 0.5, 1.0, 1.0, 2.0, 3.3, float(rng.uniform(0.05, 4))]))
                if rng.random() < 0.4:
                    pos = round(pos / unit) * unit           # an edge on a block edge
            else:
                length = total * (0.02 + 0.5 * rng.random())
            s.op(*_clip_args(rng, t, si, resample_ok[si], pos, length, frames))
            pos += length + (0.0 if rng.random() < (0.7 if kind == "dense" else 0.3) else total * 0.1 * rng.random())
    if start:
        s.op("seek", start)
    s.op("play")
    if kind in ("static", "dense"):
        s.op("run", n_blocks)
        s.op("clips")
        return s
    done = 0
    ntr = n_tracks
    while done < n_blocks:
        k = int(rng.integers(1, 4))
        s.op("run", k)
        done += k
        ph = start + done * unit                    # (about: tempo changes move it; good enough to aim edits)
        if kind == "controls":
            op = int(rng.integers(0, 11))
            t = int(rng.integers(0, ntr))
            if op == 10:     # the audio back end comes back with another device rate, in mid-play (set_audio_channel_config)
                s.op("rate", int(rng.choice([44100, 48000, 96000, 22050])))
            elif op == 0:
                s.op("vol", t, float(np.float32(rng.choice([rng.uniform(-40, 6), -72.0, -71.99, -100.0, 0.0, 6.0]))))
            elif op == 1:
                s.op("pan", t, float(np.float32(rng.choice([rng.uniform(-1, 1), -1.0, 1.0, 0.0]))))
            elif op == 2:
                s.op("mute", t, int(rng.integers(0, 2)))
            elif op == 3:
                s.op("stop"); s.op("play")
            elif op == 4:
                s.op("seek", float(rng.uniform(0, total)))
                if rng.random() < 0.5:
                    s.op("stop"); s.op("play")
            elif op == 5:
                s.op("bpm", float(rng.choice([120.0, 90.0, 133.3, 200.0])))
            elif op == 6:
                s.op("solo", t)
            elif op == 7 and ntr > 1:
                s.op("movetrack", t, int(rng.integers(0, ntr)))
            elif op == 8 and ntr > 1:
                s.op("deltrack", t); ntr -= 1
            else:
                s.op("stop"); s.op("run", 1); done += 1; s.op("play")
        else:
            op = int(rng.integers(0, 7))
            t = int(rng.integers(0, ntr))
            for _ in range(int(rng.integers(0, 3))):      # what add / move / resize / delete_region would hand to reserve_track_region:
                a = float(rng.uniform(-0.1, 1.2)) * total   # Track::query_clip_by_range on ranges that fall in gaps, span clips,
                b = a + float(rng.choice([rng.uniform(0, 0.5) * total, unit * rng.uniform(0, 2), -unit * rng.uniform(0, 2), 0.0]))   # are empty or INVERTED (Q11)
                s.op("query", int(rng.integers(0, ntr)), a, b)
            if op <= 1:      # delete a clip (index 0..3: often the one that sounds; out of range -> status 2 on both sides)
                s.op("delclip", t, int(rng.integers(0, 4)))
            elif op == 2:
                s.op("gain", t, int(rng.integers(0, 4)), float(np.float32(rng.uniform(0.0, 1.5))))
            elif op == 3:    # a clip far ahead: free space, and the pool chunk of whatever was destroyed last (Q10)
                si = int(rng.integers(0, n_tracks))
                mn = ph + total * float(rng.uniform(0.3, 0.6))
                s.op(*_clip_args(rng, t, si, resample_ok[si], mn, unit * float(rng.uniform(0.3, 4)), s.samples[si][3]))
            elif op == 4:    # a clip somewhere: mostly refused (lands on clips), sometimes a gap
                si = int(rng.integers(0, n_tracks))
                mn = float(rng.uniform(0, total))
                s.op(*_clip_args(rng, t, si, resample_ok[si], mn, unit * float(rng.uniform(0.1, 2)), s.samples[si][3]))
            elif op == 5:    # move: taken only when the destination is free of every clip, the moved one included
                s.op("move", t, int(rng.integers(0, 4)), float(rng.choice([-1, 1])) * total * float(rng.uniform(0.5, 3)))
            else:
                s.op("seek", float(rng.uniform(0, total))); s.op("stop"); s.op("play")
        s.op("clips")
    return s
