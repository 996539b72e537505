"""Seeded random sessions and clip-edit scripts shared by the GPU parity tests (product engine vs oracle) and the
CPU-side host tests (the product's host code + sequencer source vs oracle)."""
import os

import numpy as np

import oracle_ffi as O
from whitebox_amd import synth


def random_session(seed):
    """A small session with everything the sequencer and the sampler can meet at once: several clips per track at
    random beat positions (touching, short, starting/ending mid-block, beyond the sample's end), random start
    offsets, stretch speeds on both sides of 1, 44.1/48/96 kHz sources, all PCM formats, mono and stereo, mutes,
    random gains, sub-buses, odd block sizes."""
    rng = np.random.default_rng(seed)
    n_tracks = int(rng.integers(1, 28))
    block = int(rng.choice([64, 128, 256, 512]))
    # (from 8 blocks on a render takes the batch path and its instances, below the callback path's; WBX_FUZZ_MAX_BLOCKS: soak runs)
    n_blocks = int(rng.integers(2, int(os.environ.get("WBX_FUZZ_MAX_BLOCKS", "12")) + 1))
    sr = 48000
    bpm = float(rng.choice([120.0, 97.0, 140.5]))
    beat_frames = sr * 60.0 / bpm
    total_beats = n_blocks * block / beat_frames
    samples, clips = [], []
    for t in range(n_tracks):
        fmt = str(rng.choice(["f32", "f32", "i16", "i24", "i32"]))
        samples.append(synth.SampleSpec(seed_track=t, channels=int(rng.integers(1, 3)), rate=int(rng.choice([44100, 48000, 96000])),
                                        frames=int(rng.integers(300, 5000)), fmt=fmt, amp=0.2 if fmt == "f32" else 1.0))
        pos = -0.2 * total_beats * rng.random() if rng.random() < 0.3 else total_beats * rng.random() * 0.3
        for _ in range(int(rng.integers(0, 4))):
            length = total_beats * (0.02 + 0.5 * rng.random())
            speed = float(rng.choice([1.0, 1.0, 0.5, 0.8, 0.91875, 0.999, 1.0625, 1.9, 0.3]))
            clips.append(synth.ClipSpec(track=t, min_beat=float(pos), max_beat=float(pos + length),
                                        start_offset=float(rng.integers(0, 400)), speed=speed, gain=float(rng.choice([1.0, 0.5, 1.3]))))
            pos += length + (0.0 if rng.random() < 0.3 else total_beats * 0.1 * rng.random())   # touching or a gap
    n_buses = int(rng.choice([0, 0, 3]))
    return synth.SessionSpec(name=f"fuzz{seed}", n_tracks=n_tracks, seed=0xF0220000 + seed, samples=samples, clips=clips,
                             volumes_db=[float(rng.uniform(-30, 3)) for _ in range(n_tracks)],
                             pans=[float(rng.uniform(-1, 1)) for _ in range(n_tracks)],
                             mutes=[bool(rng.random() < 0.1) for _ in range(n_tracks)],
                             n_buses=n_buses, track_bus=[int(rng.integers(-1, n_buses)) for _ in range(n_tracks)] if n_buses else None,
                             bpm=bpm, sample_rate=sr, block=block, channels=int(rng.choice([1, 2, 2])),
                             playhead_start=float(rng.choice([0.0, 0.0, total_beats * 0.1]))), n_blocks



def random_masked_session(seed, integer_unity=False, lean16=False, everything=False):
    """Sessions the masked-row path of the mix kernel takes: fp32 clips only, 44.1 / 48 kHz sources, stretch speeds on the
    unity and 5-sample-window paths, stereo 512-frame (or mono / stereo 1024-frame) blocks — with everything a clip list
    can do to a block: touching clips, gaps, clips shorter than a block (three and more stream calls: those still go to
    the pre-render pass), clips that outlast their audio, random start offsets and gains, mutes, sub-buses, more tracks
    than one staged chunk.  `integer_unity`: the second kind of session that path takes — clips of every storage format
    (16 / 24 / 32-bit PCM, fp32; some sessions one format only), all recorded at the session rate and played at speed 1.
    `lean16`: the third kind — 16-bit PCM only, 44.1 / 48 kHz sources, the same stretch speeds as the fp32 sessions.
    `everything`: what the everything family of the mix kernel takes — every storage format (some sessions 24-bit only,
    or 16- and 24-bit), 44.1 / 48 / 96 kHz sources, stretch speeds below and above 1 (window rows of every format,
    per-frame-tap rows), always at least one clip that forces that family."""
    rng = np.random.default_rng((0xE7E0000 if everything else 0x16B0000 if lean16 else 0x1C70000 if integer_unity else 0xA5C0) + seed)
    fmts = ["i16"] if lean16 else ["f32"]
    if integer_unity:
        fmts = [["i16"], ["i24"], ["i32"], ["i16", "i24", "i32", "f32"], ["i16", "f32"]][int(rng.integers(0, 5))]
    if everything:
        fmts = [["i24"], ["i16", "i24"], ["i16", "i24", "i32", "f32"], ["f32"], ["i32", "f32"]][int(rng.integers(0, 5))]
    n_tracks = int(rng.choice([3, 17, 40, 130, 200]))
    block, channels = [(512, 2), (512, 2), (1024, 2), (1024, 1), (256, 2), (128, 2), (512, 1), (256, 1)][int(rng.integers(0, 8))]
    # (from 8 blocks on a render takes the batch path and its instances, below the callback path's; WBX_FUZZ_MAX_BLOCKS: soak runs)
    n_blocks = int(rng.integers(2, int(os.environ.get("WBX_FUZZ_MAX_BLOCKS", "12")) + 1))
    sr = 48000
    bpm = float(rng.choice([120.0, 97.0, 140.5]))
    beat_frames = sr * 60.0 / bpm
    total_beats = n_blocks * block / beat_frames
    samples, clips = [], []
    for t in range(n_tracks):
        samples.append(synth.SampleSpec(seed_track=t, channels=int(rng.integers(1, 3)),
                                        rate=sr if integer_unity else int(rng.choice([44100, 48000, 96000] if everything else [44100, 48000])),
                                        frames=int(rng.integers(600, 9000)), fmt=str(rng.choice(fmts)), amp=0.05))
        pos = -0.2 * total_beats * rng.random() if rng.random() < 0.3 else total_beats * rng.random() * 0.3
        for _ in range(int(rng.integers(0, 6))):
            length = total_beats * (0.02 + 0.45 * rng.random())
            speed = 1.0 if integer_unity else float(rng.choice([1.0, 1.0, 0.5, 0.8, 0.91875, 0.999, 0.3, 0.67]))
            if everything:
                speed = float(rng.choice([1.0, 1.0, 0.5, 0.8, 0.999, 0.3, 1.25, 2.0, 1.0005]))
            clips.append(synth.ClipSpec(track=t, min_beat=float(pos), max_beat=float(pos + length),
                                        start_offset=float(rng.integers(0, 400)), speed=speed, gain=float(rng.choice([1.0, 0.5, 1.3]))))
            pos += length + (0.0 if rng.random() < 0.5 else total_beats * 0.1 * rng.random())   # touching or a gap
    if everything and fmts == ["f32"]:   # an fp32-only session takes the everything family through a per-frame-tap clip
        samples[0].rate = 96000
        clips.append(synth.ClipSpec(track=0, min_beat=total_beats * 0.9, max_beat=total_beats * 0.97, start_offset=3.0, speed=1.0))
    n_buses = int(rng.choice([0, 0, 3]))
    return synth.SessionSpec(name=f"{'efuzz' if everything else 'sfuzz' if lean16 else 'ifuzz' if integer_unity else 'mfuzz'}{seed}", n_tracks=n_tracks, seed=0xF0330000 + seed, samples=samples, clips=clips,
                             volumes_db=[float(rng.uniform(-30, 3)) for _ in range(n_tracks)],
                             pans=[float(rng.uniform(-1, 1)) for _ in range(n_tracks)],
                             mutes=[bool(rng.random() < 0.1) for _ in range(n_tracks)],
                             n_buses=n_buses, track_bus=[int(rng.integers(-1, n_buses)) for _ in range(n_tracks)] if n_buses else None,
                             bpm=bpm, sample_rate=sr, block=block, channels=channels,
                             playhead_start=float(rng.choice([0.0, 0.0, total_beats * 0.1]))), n_blocks


def clip_rows(clips):
    return [(O.f64_bits(a), O.f64_bits(b), O.f64_bits(c), O.f64_bits(d), O.f32_bits(g), s) for (a, b, c, d, g, s) in clips]


def edit_session_spec(seed):
    """6 tracks, no clips yet; seed 2024 is the original all-fp32 512-frame script, the others also draw the block
    size, the storage formats and the sample rates."""
    n_tracks = 6
    block = 512 if seed == 2024 else int(np.random.default_rng(seed + 7).choice([512, 256, 128, 64]))
    spec = synth.make_session("edits", n_tracks, n_blocks=40, seed=0xED17, amp=0.05, block=block)
    spec.clips = []
    for i, s in enumerate(spec.samples):
        s.frames = 40000
        if seed != 2024:
            r2 = np.random.default_rng(seed * 31 + i)
            s.fmt = str(r2.choice(["f32", "f32", "i16", "i24"]))
            s.rate = int(r2.choice([48000, 48000, 44100, 96000]))
            s.amp = 0.05 if s.fmt == "f32" else 1.0
    if seed != 2024:
        for t in range(n_tracks):
            spec.volumes_db[t] = -30.0
    return spec


def run_edit_script(seed, spec, e, eng, on_block, steps=24, sounding_bias=True):
    """The same random edits through the oracle `e` and an engine `eng` with the reference-shaped method names
    (whitebox_amd.engine.Engine or tests/host_sim.HostSimEngine); after every edit the sorted clip lists must agree
    bit for bit, then on_block(step, op) renders one block on both and compares what it can.  Returns the number of
    (block, track) pairs that streamed through a destroyed clip (quirk Q10)."""
    rng = np.random.default_rng(seed)
    rng2 = np.random.default_rng(seed ^ 0xD46)
    n_tracks, beat = spec.n_tracks, 24000.0

    def both(fn_o, fn_p):
        ro = fn_o()
        fn_p()
        return ro

    # overlapping adds: the new clip trims / splits / deletes what it covers (reserve_track_region)
    for t in range(n_tracks):
        for _ in range(6):
            mn = float(rng.uniform(0, 8000)) / beat
            mx = mn + float(rng.uniform(300, 5000)) / beat
            so = float(rng.integers(0, 500))
            sp = float(rng.choice([1.0, 1.0, 0.5, 1.5]))
            g = float(np.float32(rng.choice([1.0, 0.5, 0.25])))
            smp = int(rng.integers(0, n_tracks))
            both(lambda: e.add_audio_clip(t, mn, mx, so, smp, sp, g),
                 lambda: eng.add_audio_clip(eng.tracks[t], "c", mn, mx, so, smp, sp, g))
        assert clip_rows(eng.clips(eng.tracks[t])) == clip_rows(e.clips(t)), t
    e.play()
    eng.play()
    hits = 0
    for step in range(steps):
        # one edit per step on a random track, then one block
        t = int(rng.integers(0, n_tracks))
        n = len(e.clips(t))
        op = int(rng.integers(0, 6))
        # A third of the steps aim at the clip that is SOUNDING (quirk Q10: an edit destroys it, no event follows, the
        # reference streams on through the freed Clip whose gain now reads 0.0f — or the gain of a newer clip that took
        # the pool chunk over).  Ops 6-8 are drawn from a generator of their own so that the scripts of earlier rounds
        # keep their op sequence where no such op is drawn.
        ph = e.playhead
        if sounding_bias and rng2.random() < 0.34:
            playing = [tt for tt in range(n_tracks) if e.sounding(tt)]
            if playing:
                t = int(rng2.choice(playing))
                n = len(e.clips(t))
                op = int(rng2.integers(6, 10))
        if op == 6:      # a new clip over the playhead: reserve_track_region trims / deletes what is playing
            mn = max(0.0, ph - float(rng2.uniform(0, 3000)) / beat)
            mx = ph + float(rng2.uniform(600, 4000)) / beat
            g = float(np.float32(rng2.choice([1.0, 0.5, 0.25])))
            both(lambda: e.add_audio_clip(t, mn, mx, 0.0, t, 1.0, g),
                 lambda: eng.add_audio_clip(eng.tracks[t], "c", mn, mx, 0.0, t, 1.0, g))
        elif op == 7:    # the next clip's left edge dragged back across the playhead (no shift / stretch: no state change)
            cl = e.clips(t)
            nxt = [i for i, c in enumerate(cl) if c[0] > ph]
            if nxt:
                i = nxt[0]
                rel = -(cl[i][0] - ph) - float(rng2.uniform(100, 3000)) / beat
                both(lambda: e.resize_clip(t, i, rel, 0.0, 1.0 / 96.0, True, False, False),
                     lambda: eng.resize_clip(eng.tracks[t], i, rel, 0.0, 1.0 / 96.0, True, False, False))
            else:
                mn, mx = max(0.0, ph - 2000.0 / beat), ph + 2500.0 / beat
                both(lambda: e.add_audio_clip(t, mn, mx, 0.0, t, 1.0, 1.0),
                     lambda: eng.add_audio_clip(eng.tracks[t], "c", mn, mx, 0.0, t, 1.0, 1.0))
        elif op == 8:    # a region delete around the playhead
            mn = max(0.0, ph - float(rng2.uniform(0, 6000)) / beat)
            mx = ph + float(rng2.uniform(100, 6000)) / beat
            both(lambda: e.delete_region(t, mn, mx), lambda: eng.delete_region(eng.tracks[t], mn, mx))
        elif op == 9:    # a clip far ahead: it takes over the pool chunk of whatever the track destroyed last
            mn = ph + float(rng2.uniform(6000, 9000)) / beat
            mx = mn + float(rng2.uniform(300, 2000)) / beat
            g = float(np.float32(rng2.choice([0.75, 0.5, 1.25])))
            both(lambda: e.add_audio_clip(t, mn, mx, 0.0, t, 1.0, g),
                 lambda: eng.add_audio_clip(eng.tracks[t], "c", mn, mx, 0.0, t, 1.0, g))
        elif n and op == 0:
            i, rel = int(rng.integers(0, n)), float(rng.normal(0, 1500)) / beat
            both(lambda: e.move_clip(t, i, rel), lambda: eng.move_clip(eng.tracks[t], i, rel))
        elif n and op == 1:
            i, rel = int(rng.integers(0, n)), float(rng.normal(0, 800)) / beat
            left, shift, stretch = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            both(lambda: e.resize_clip(t, i, rel, 0.0, 1.0 / 96.0, left, shift, stretch),
                 lambda: eng.resize_clip(eng.tracks[t], i, rel, 0.0, 1.0 / 96.0, left, shift, stretch))
        elif n and op == 2:
            i = int(rng.integers(0, n))
            both(lambda: e.delete_clip(t, i), lambda: eng.delete_clip(eng.tracks[t], i))
        elif n and op == 3:
            i, g = int(rng.integers(0, n)), float(np.float32(rng.uniform(0.1, 1.5)))
            both(lambda: e.set_clip_gain(t, i, g), lambda: eng.set_clip_gain(eng.tracks[t], i, g))
        elif op == 4:
            mn = float(rng.uniform(0, 12000)) / beat
            mx = mn + float(rng.uniform(100, 3000)) / beat
            both(lambda: e.delete_region(t, mn, mx), lambda: eng.delete_region(eng.tracks[t], mn, mx))
        else:
            mn = float(rng.uniform(0, 14000)) / beat
            mx = mn + float(rng.uniform(200, 4000)) / beat
            both(lambda: e.add_audio_clip(t, mn, mx, 0.0, t, 1.0, 1.0),
                 lambda: eng.add_audio_clip(eng.tracks[t], "c", mn, mx, 0.0, t, 1.0, 1.0))
        for tt in range(n_tracks):
            assert clip_rows(eng.clips(eng.tracks[tt])) == clip_rows(e.clips(tt)), (step, op, tt)
        on_block(step, op)
        hits += sum(1 for tt in range(n_tracks) if e.dangling(tt))
    return hits


# ----------------------------------------------------------------------------------------------------------------------
# "wild" sessions (round 5): the corners of what the reference's API accepts, all at once
# ----------------------------------------------------------------------------------------------------------------------
WILD_RATES = [22050, 32000, 44100, 48000, 88200, 96000, 192000]
WILD_BLOCKS = [32, 64, 100, 128, 440, 512, 1000, 2048]      # (the product takes multiples of 4 frames)


def _wild_clip(rng, t, samples, unit, total, around=None):
    """one clip of track t: sub-frame, sub-block and many-block lengths, edges snapped to block edges now and then, start
    offsets beyond the sample's end, crawling (0.01) and racing (8) stretch factors, NEGATIVE ones (what a stretch-shrink past
    the sample's own length leaves behind — quirk Q12), gains of exactly 0"""
    mn = float(rng.uniform(0, total)) if around is None else max(0.0, around + float(rng.normal(0, 3 * unit)))
    ln = float(rng.choice([1e-6, 0.01, 0.5, 1.0, 1.0, 3.3, 10.0])) * unit * float(rng.uniform(0.5, 1.5))
    if rng.random() < 0.25:
        mn, ln = round(mn / unit) * unit, max(1, round(ln / unit)) * unit
    s = int(rng.integers(0, len(samples)))
    so = float(rng.choice([0.0, 0.0, float(rng.integers(0, samples[s].frames + 50)), float(rng.uniform(0, 100))]))
    sp = float(rng.choice([1.0, 1.0, 0.5, 2.0, 0.01, 8.0, float(rng.uniform(0.2, 3.0)), -float(rng.uniform(0.05, 2.0))]))
    g = float(np.float32(rng.choice([1.0, 0.0, 0.5, 2.0, float(rng.uniform(0, 2))])))
    return synth.ClipSpec(track=t, min_beat=mn, max_beat=mn + ln, start_offset=so, speed=sp, gain=g, sample=s)


def wild_session(seed):
    """1-5 tracks over 1-4 shared samples of 5 … 200 000 frames (every storage format, 8 … 192 kHz), session rates 22.05 …
    192 kHz, blocks of 32 … 2048 frames (100, 440 and 1000 among them), any tempo, up to 13 clips per track laid out at random
    (overlapping adds trim / split / delete each other), a playhead that may start anywhere.  Returns (spec, n_blocks)."""
    rng = np.random.default_rng(seed * 7907 + 5)
    sr = int(rng.choice(WILD_RATES))
    block = int(rng.choice(WILD_BLOCKS))
    bpm = float(rng.uniform(30.0, 300.0)) if rng.random() < 0.7 else float(rng.choice([120.0, 60.0, 90.0, 128.0]))
    nt, ns = int(rng.integers(1, 6)), int(rng.integers(1, 5))
    samples = []
    for i in range(ns):
        fmt = str(rng.choice(["f32", "i16", "i24", "i32"]))
        rate = int(rng.choice([8000, 11025, 22050, 44100, 48000, 96000, sr, sr]))
        frames = int(rng.choice([5, 17, 300, 5000, 50000, 200000]))
        samples.append(synth.SampleSpec(i, int(rng.integers(1, 3)), rate, frames, fmt, 0.2 if fmt == "f32" else 1.0))
    unit = block / (sr * 60.0 / bpm)            # one block, in beats
    n_blocks = int(rng.integers(12, 40))
    total = n_blocks * unit
    clips = [_wild_clip(rng, t, samples, unit, total) for t in range(nt) for _ in range(int(rng.integers(0, 14)))]
    spec = synth.SessionSpec(f"wild{seed}", nt, 0xBEEF00 + seed, samples, clips,
                             [float(np.float32(rng.uniform(-80, 6))) for _ in range(nt)],
                             [float(np.float32(rng.uniform(-1, 1))) for _ in range(nt)],
                             [bool(rng.random() < 0.1) for _ in range(nt)], bpm=bpm, sample_rate=sr, block=block, channels=int(rng.choice([2, 2, 1])),
                             playhead_start=float(rng.choice([0.0, 0.0, float(rng.uniform(0, total))])))
    return spec, n_blocks


def run_wild_script(seed, spec, n_blocks, e, eng, on_block, edits=True, pieces=False):
    """Plays the session block by block on the oracle `e` and an engine `eng` (whitebox_amd.engine.Engine or
    tests/host_sim.HostSimEngine); with `edits`, 0-2 operations of everything a host can do go between two blocks: parameter
    messages, stop / play, seeks, tempo changes, clip gain / delete / move / resize (shift, stretch — shrinks that leave a
    NEGATIVE speed behind included), adds around the playhead, region deletes, track moves and deletes.  The clip lists must
    agree bit for bit after every operation; on_block(b, trail) renders and compares one block — with `pieces`,
    on_block(b, trail, k) renders k = 1 … 16 blocks from block b on (the operations then fall between pieces; from 8 blocks on
    a piece takes the batch path) and counts the Q12 calls itself.  Returns the number of stream calls that ran a tap index
    below zero (quirk Q12)."""
    rng = np.random.default_rng(seed * 104729 + 71)
    unit = spec.block / (spec.sample_rate * 60.0 / spec.bpm)
    total = n_blocks * unit
    nt = spec.n_tracks
    trail, q12 = [], 0
    for t in range(nt):
        assert clip_rows(eng.clips(eng.tracks[t])) == clip_rows(e.clips(t)), ("initial clip list", t)
    e.play()
    eng.play()
    rng_k = np.random.default_rng(seed * 31337 + 3)
    b = -1
    while b + 1 < n_blocks:
        b += 1
        for _ in range(int(rng.integers(0, 3)) if edits else 0):
            if nt == 0:
                break
            t = int(rng.integers(0, nt))
            cl = e.clips(t)
            n, op, ph = len(cl), int(rng.integers(0, 14)), e.playhead
            T = eng.tracks[t]
            trail.append((b, op, t))
            if op == 0:
                v = float(np.float32(rng.uniform(-80, 6))); e.set_volume(t, v); T.set_volume(v)
            elif op == 1:
                v = float(np.float32(rng.uniform(-1, 1))); e.set_pan(t, v); T.set_pan(v)
            elif op == 2:
                m = bool(rng.integers(0, 2)); e.set_mute(t, m); T.set_mute(m)
            elif op == 3:
                e.stop(); eng.stop()
                if rng.random() < 0.8:
                    e.play(); eng.play()
            elif op == 4:
                bt = float(rng.uniform(0, total)); e.set_playhead(bt); eng.set_playhead_position(bt)
            elif op == 5:
                nb = float(rng.uniform(30, 300)); e.set_bpm(nb); eng.set_bpm(nb)
            elif op == 6 and n:
                i, g = int(rng.integers(0, n)), float(np.float32(rng.uniform(0, 2)))
                e.set_clip_gain(t, i, g); eng.set_clip_gain(T, i, g)
            elif op == 7 and n:
                i = int(rng.integers(0, n)); e.delete_clip(t, i); eng.delete_clip(T, i)
            elif op == 8 and n:
                i, rel = int(rng.integers(0, n)), float(rng.normal(0, 4 * unit))
                e.move_clip(t, i, rel); eng.move_clip(T, i, rel)
            elif op == 9 and n:
                i, rel = int(rng.integers(0, n)), float(rng.normal(0, 2 * unit))
                left, shift, stretch = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
                span = cl[i][1] - cl[i][0]      # the new edge stays on its own side of the other one (the inverted range is Q11)
                rel = min(rel, span * 0.9) if left else max(rel, -span * 0.9)
                e.resize_clip(t, i, rel, 0.0, 1.0 / 96.0, left, shift, stretch); eng.resize_clip(T, i, rel, 0.0, 1.0 / 96.0, left, shift, stretch)
            elif op == 10:
                c = _wild_clip(rng, t, spec.samples, unit, total, around=ph)
                e.add_audio_clip(t, c.min_beat, c.max_beat, c.start_offset, c.sample, c.speed, c.gain)
                eng.add_audio_clip(T, "w", c.min_beat, c.max_beat, c.start_offset, c.sample, c.speed, c.gain)
            elif op == 11:
                mn = max(0.0, ph + float(rng.normal(0, 3 * unit))); mx = mn + float(rng.uniform(0.1, 5)) * unit
                e.delete_region(t, mn, mx); eng.delete_region(T, mn, mx)
            elif op == 12 and nt > 1 and rng.random() < 0.3:
                a_, b_ = int(rng.integers(0, nt)), int(rng.integers(0, nt)); e.move_track(a_, b_); eng.move_track(a_, b_)
            elif op == 13 and nt > 1 and rng.random() < 0.15:
                sl = int(rng.integers(0, nt)); e.delete_track(sl); eng.delete_track(sl); nt -= 1
            for tt in range(nt):
                assert clip_rows(eng.clips(eng.tracks[tt])) == clip_rows(e.clips(tt)), ("clip list", trail[-1], tt)
        if pieces:
            k = min(int(rng_k.integers(1, 17)), n_blocks - b)
            q12 += on_block(b, trail, k) or 0
            b += k - 1
            continue
        on_block(b, trail)
        q12 += q12_calls(e.seglog())
    return q12


def q12_calls(seglog):
    """stream calls of a block's log that run a tap index below zero (quirk Q12)"""
    return sum(1 for sg in seglog if sg[4] < 0 and sg[2] > 0 and sg[3] + (sg[2] - 1) * sg[4] <= -1.0)
