#!/usr/bin/env python3
"""bench.py — headline benchmark of the whitebox mix path on MI355X.

  python bench.py --gpus N --steps K --warmup W
      N>1: either launched one rank per GPU by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
      environment), or run plainly — bench.py then starts the N ranks itself.

A "step" is one device pass of the hot path (clip sequencer + per-track render/gain/pan/resample +
peaks + group/bus/master sum + clamp) over one batch of `--blocks` consecutive 512-frame blocks (default 2048: the
tracks of every block are added in the reference's strictly sequential order) of a synthetic session that is already
resident in HBM.  After the timed loop the head of one more step is checked against the CPU oracle (`verify`).  The default workload is BASELINE.json configs[2]
("4096 stereo tracks, gain+pan + linear clip resample (44.1->48 kHz), 1 MI355X") — the configuration the
metric "…4096 tracks @ 512-frame blocks" is quoted on.  With N GPUs every rank mixes its own 4096 tracks
(weak scaling = configs[4]: 32768 tracks sharded 8-way) and the un-clamped partial masters are reduced
to rank 0 with one RCCL reduce per step, then clamped there — all of it inside libwbx.so (wbx_dist_*), no torch.

Prints ONE JSON line on rank 0.  `value` = 4096-track-equivalent stereo frames mixed per second over the
whole job = (total tracks / 4096) x master frames / wall time; at N=1 it is exactly master frames/s.
"""
import argparse
import ctypes as C
import gc
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
F = 512
SR = 48000

WORKLOADS = {
    # name: (description, src_rate, n_buses, clip storage format)
    "c2": ("configs[1]: 256 stereo tracks, per-track gain+pan, 512-frame blocks", 48000, 0, "f32"),
    "c3": ("configs[2]: 4096 stereo tracks, gain+pan + linear clip resample 44.1k->48k, 512-frame blocks", 44100, 0, "f32"),
    "c4": ("configs[3]: 4096 stereo tracks into 64 sub-buses + master sum, 512-frame blocks", 48000, 64, "f32"),
    "u4096": ("config 4's input side: 4096 stereo tracks at the session rate straight into the master (no buses), 512-frame blocks",
              48000, 0, "f32"),
    "i16": ("next-2 (SURVEY 8f): 4096 stereo 16-bit PCM tracks, gain+pan, unity rate, 512-frame blocks", 48000, 0, "i16"),
    "d96": ("4096 stereo 96 kHz tracks played in the 48 kHz session (playback speed 2, per-frame taps), 512-frame blocks",
            96000, 0, "f32"),
    "i16r": ("4096 stereo 16-bit 44.1 kHz tracks resampled into the 48 kHz session (per-frame taps), 512-frame blocks",
             44100, 0, "i16"),
    "i24r": ("4096 stereo 24-bit 44.1 kHz tracks (32-bit containers) resampled into the 48 kHz session, 512-frame blocks",
             44100, 0, "i24"),
    "mixfmt": ("4096 stereo tracks at the session rate, storage format cycling fp32 / 16-bit / 24-bit per track, 512-frame blocks",
               48000, 0, "mix3"),
    "mixr": ("4096 stereo tracks, alternating 16-bit 44.1 kHz clips (resampled) and 24-bit 48 kHz clips, 512-frame blocks",
             48000, 0, "mixr"),
}
SEEDS = {"c2": 2, "c3": 3, "c4": 4, "u4096": 11, "i16": 5, "d96": 6, "i16r": 7, "i24r": 8, "mixfmt": 9, "mixr": 10}
# bytes per stored sample (the mixed workloads: mean over their tracks, rate ratio folded in for mixr)
FMT_BYTES = {"f32": 4, "i16": 2, "i24": 4, "i32": 4, "mix3": 10.0 / 3.0, "mixr": (2 * 0.91875 + 4) / 2}


def algorithmic_bytes_per_block(n_tracks: int, src_rate: int, channels: int = 2, fmt: str = "f32") -> float:
    """per master frame read N*C*B*r bytes of clip audio (B = bytes per stored sample, r =
    src/dst rate) and write C*4 bytes of master; per block additionally N*C*4 B of peaks and N*32 B of
    segment/gain tables."""
    r = src_rate / SR
    return F * (n_tracks * channels * FMT_BYTES[fmt] * r + channels * 4) + n_tracks * channels * 4 + n_tracks * 32


def track_layout(workload, n_tracks, rank, world, session_blocks, clip_blocks=0.0):
    """The synthetic session, one entry per local track: what both the device engine and the oracle are built from.
    -> (amp, [(global_track, format, src_rate, volume_dB, pan, bus, [(min_beat, max_beat, start_offset), ...]), ...])"""
    from whitebox_amd import synth
    _, src_rate, n_buses, fmt = WORKLOADS[workload]
    seed = 0x5EED0000 + SEEDS[workload]
    total_tracks = n_tracks * world
    amp = synth.default_amp(total_tracks)
    beat_frames = SR * 60.0 / 120.0
    per_bus = max(1, n_tracks // n_buses) if n_buses else 0
    out = []
    for t in range(n_tracks):
        gt = rank * n_tracks + t                      # global track index keys the generator and the parameters
        tfmt = ("f32", "i16", "i24")[gt % 3] if fmt == "mix3" else ("i16", "i24")[gt % 2] if fmt == "mixr" else fmt
        trate = 44100 if (fmt == "mixr" and gt % 2 == 0) else src_rate
        v, p = synth.track_params(seed, gt)
        if tfmt != "f32":
            v = v - 20.0 * math.log10(0.25 / amp)     # integer clips are full scale: the session level goes into the faders
        bus = min(t // per_bus, n_buses - 1) if n_buses else -1
        clips = []
        if clip_blocks <= 0.0:
            clips.append((0.0, (session_blocks + 1) * F / beat_frames, 0.0))
        else:
            # side measurement: the track is cut into back-to-back clips of clip_blocks blocks (each reading on from
            # where the previous one stopped), staggered per track
            L = clip_blocks * F
            pos = -((t * 37) % 512) / 512.0 * L
            while pos < (session_blocks + 1) * F:
                a, b = max(pos, 0.0), pos + L
                clips.append((a / beat_frames, b / beat_frames, a * (src_rate / SR)))
                pos = b
        out.append((gt, tfmt, trate, float(v), float(p), bus, clips))
    return seed, amp, out


def build_device_session(W, synth, workload, n_tracks, blocks, session_blocks, rank, world, group_size, clip_blocks=0.0):
    from whitebox_amd.engine import Engine
    desc, src_rate, n_buses, fmt = WORKLOADS[workload]
    # (A/B aid WBX_MASKED_ROWS=0 sends every clip boundary through the pre-render pass: its segment pool must hold them)
    pre_render = clip_blocks and os.environ.get("WBX_MASKED_ROWS") == "0"
    eng = Engine(n_tracks, F, SR, 2, max_blocks=blocks, group_size=group_size, device=rank_device(rank),
                 max_segments=4 * n_tracks * blocks if pre_render else 0)
    eng.set_bpm(120.0)
    if n_buses:
        eng.set_buses(n_buses)
    seed, amp, tracks = track_layout(workload, n_tracks, rank, world, session_blocks, clip_blocks)
    frames = int(math.ceil((session_blocks + 2) * F * (src_rate / SR))) + 64
    for (gt, tfmt, trate, v, p, bus, clips) in tracks:
        sid = eng.add_sample_synth(tfmt, 2, trate, frames, seed, gt, amp)
        tr = eng.add_track(f"t{gt}")
        tr.set_volume(v)
        tr.set_pan(p)
        if n_buses:
            tr.set_bus(bus)
        for (a, b, off) in clips:
            eng.add_audio_clip(tr, "clip", a, b, off, sid, 1.0, 1.0)
    return eng, seed, amp


def rank_device(rank):
    dev = int(os.environ.get("LOCAL_RANK", rank))
    if os.environ.get("WBX_SHARE_DEVICE") == "1":
        import whitebox_amd as W
        dev %= max(1, W.lib().wbx_device_count())
    return dev


def build_oracle_session(workload, n_tracks, world, sample_blocks, clip_blocks=0.0, session_blocks=None):
    """The same session (all `world` shards of it, in global track order) in the CPU oracle, with clip audio for the
    first sample_blocks (+2) blocks only — identical to the head of the resident device session, whose generator is
    keyed by (track, channel, frame).  TEST INFRASTRUCTURE: used by the verify step and the cpu_baseline leg only."""
    import oracle_ffi as O
    from whitebox_amd import synth
    _, src_rate, n_buses, fmt = WORKLOADS[workload]
    L = O.lib()
    L.wbo_synth_f32.argtypes = [O.c_f32p, C.c_size_t, C.c_uint64, C.c_float, C.c_size_t]
    e = O.OracleEngine(2, F, SR)
    e.set_bpm(120.0)
    if n_buses:
        e.set_buses(n_buses * world)
    keep = []
    idx = 0
    for rank in range(world):
        seed, amp, tracks = track_layout(workload, n_tracks, rank, world, session_blocks or sample_blocks, clip_blocks)
        amp32 = np.float32(amp)
        for (gt, tfmt, trate, v, p, bus, clips) in tracks:
            frames = int(math.ceil((sample_blocks + 2) * F * (trate / SR))) + 64
            chans = []
            for c in range(2):
                if tfmt == "f32":
                    a = np.empty(frames + 16, np.float32)
                    L.wbo_synth_f32(a.ctypes.data_as(O.c_f32p), frames, int(synth.clip_key(seed, gt, c)), amp32, 16)
                elif tfmt == "i16":
                    a = np.concatenate([synth.clip_channel_i16(seed, gt, c, frames), np.zeros(16, np.int16)])
                else:
                    a = np.concatenate([synth.clip_channel_i32(seed, gt, c, frames, 24 if tfmt == "i24" else 32),
                                        np.zeros(16, np.int32)])
                chans.append(a)
            keep.append(chans)
            sid = e.add_sample(tfmt, 2, trate, frames, chans)
            e.add_track()
            e.set_volume(idx, v)
            e.set_pan(idx, p)
            if n_buses:
                e.set_bus(idx, rank * n_buses + bus)
            horizon = (sample_blocks + 1) * F / (SR * 60.0 / 120.0)
            for (a, b, off) in clips:
                if a <= horizon:
                    e.add_audio_clip(idx, a, b, off, sid, 1.0, 1.0)
            idx += 1
    e._keep_clips = keep
    return e


def cpu_baseline(workload, n_tracks, budget_s=12.0):
    """The oracle (C restatement of the reference's single-threaded Engine::process, bit-identical to the
    reference's own TUs) timed on ONE host core over a bounded sample of the same workload."""
    import oracle_ffi as O
    from whitebox_amd import synth
    _, src_rate, n_buses, fmt = WORKLOADS[workload]
    L = O.lib()
    sample_blocks = 24
    e = build_oracle_session(workload, n_tracks, 1, sample_blocks)
    keep = e._keep_clips
    seed = 0x5EED0000 + SEEDS[workload]
    amp = np.float32(synth.default_amp(n_tracks))
    out = [np.zeros(F, np.float32) for _ in range(2)]
    ptrs = O.planar_ptrs(out)
    blocks_done, elapsed, passes = 0, 0.0, 0
    while elapsed < budget_s and passes < 64:
        e.play()
        t0 = time.perf_counter()
        for _ in range(sample_blocks):
            L.wbo_engine_process(e.e, ptrs, None)
        elapsed += time.perf_counter() - t0
        e.stop()
        blocks_done += sample_blocks
        passes += 1
    e.close()
    fps = blocks_done * F / elapsed
    best = cpu_all_cores(O, L, workload, n_tracks, keep, sample_blocks, src_rate, n_buses, fmt, seed, amp)
    try:
        model = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except Exception:
        model = "unknown"
    port = {"value": fps, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n_tracks} tracks x {sample_blocks} blocks x {passes} passes ({workload}), "
                      f"{elapsed:.1f} s of CPU work, single thread like the reference's audio thread",
            "host_cpu": model, "host_cores_total": os.cpu_count(),
            "us_per_block": 1e6 * elapsed / blocks_done, "all_cores": best}
    ref = cpu_reference(workload, n_tracks, sample_blocks, budget_s)
    if ref is None:
        return port
    # the reference's OWN Engine::process timed beside it; the oracle's figure stays in the line as `port`
    ref.update({"host_cpu": model, "host_cores_total": os.cpu_count(), "all_cores": best,
                "port": {k: port[k] for k in ("value", "unit", "cores", "kind", "sample", "us_per_block")}})
    return ref


def ref_engine_stamp(exe, oracle_dir):
    """(is `exe` the build of this tree's driver source + recipe?, the tree's hash, the stamp oracle/Makefile left beside the
    executable or None, the executable's own hash or None)"""
    import hashlib
    want = hashlib.sha256(b"".join(open(os.path.join(oracle_dir, n), "rb").read()
                                   for n in ("ref_engine_driver.cpp", "Makefile"))).hexdigest()[:16]
    try:
        stamp = open(exe + ".stamp").read().strip()
    except OSError:
        stamp = None
    try:
        exe_sha = hashlib.sha256(open(exe, "rb").read()).hexdigest()[:16]
    except OSError:
        exe_sha = None
    return stamp == want and exe_sha is not None, want, stamp, exe_sha


def cpu_reference(workload, n_tracks, sample_blocks, budget_s):
    """The reference's own Engine::process (oracle/_ref/wbref_engine: Track::process_event / Track::process / Engine::process /
    Sampler::stream, cut out of the reference's sources where they lie and compiled unmodified, -O2 — oracle/Makefile) on ONE
    host core, its only thread, over the same bounded sample of the same session.  The executable is prebuilt where
    /root/reference exists and travels with the tree; nothing of /root/reference is read here.  None where it is absent or the
    workload needs what the reference does not have (sub-buses) or what the driver's generator does not make (integer PCM)."""
    import subprocess
    import tempfile
    import oracle_ffi as O
    import ref_engine as R
    exe = os.path.join(O.ORACLE_DIR, "_ref", "wbref_engine")
    _, src_rate, n_buses, fmt = WORKLOADS[workload]
    if not R.available(build=False) or n_buses or fmt != "f32":
        return None
    # the executable is a prebuilt, untracked file: say which build it is, and refuse one that was not made from the driver
    # source and the recipe of this tree (oracle/Makefile leaves their hash beside it)
    ok, want, stamp, exe_sha = ref_engine_stamp(exe, O.ORACLE_DIR)
    if not ok:
        print(f"cpu_reference: oracle/_ref/wbref_engine is stale or unstamped (stamp {stamp}, tree {want}): the port is the baseline",
              file=sys.stderr)
        return None
    seed, amp, tracks = track_layout(workload, n_tracks, 0, 1, sample_blocks)
    amp32 = float(np.float32(amp))
    lines = [f"cfg 2 {F} {SR}", "bpm 120.0"]
    for i, (gt, tfmt, trate, v, p, bus, clips) in enumerate(tracks):
        frames = int(math.ceil((sample_blocks + 2) * F * (trate / SR))) + 64       # as build_oracle_session
        lines.append(f"synth 2 {trate} {frames} {seed} {gt} {amp32.hex()}")
        lines += ["track", f"vol {i} {float(np.float32(v))!r}", f"pan {i} {float(np.float32(p))!r}"]
        for (a, b, off) in clips:
            lines.append(f"clip {i} {float(a).hex()} {float(b).hex()} {float(off).hex()} {i} {1.0.hex()} {float(np.float32(1.0)).hex()}")
    lines.append(f"bench {sample_blocks} {budget_s} 64")
    try:
        with tempfile.TemporaryDirectory() as d:
            sp, dp, rp = (os.path.join(d, n) for n in ("script.txt", "data.bin", "result.bin"))
            open(sp, "w").write("\n".join(lines) + "\n")
            open(dp, "wb").close()
            r = subprocess.run([exe, sp, dp, rp], timeout=budget_s * 4 + 120, capture_output=True)
            if r.returncode != 0:
                return None
            rec = [x for x in R.parse_results(open(rp, "rb").read(), 2, F) if x[0] == "bench"][0][1]
    except Exception as ex:                                            # a baseline, not the product: never fails the bench
        print(f"cpu_reference: {ex!r}", file=sys.stderr)
        return None
    # what was timed is this session: the first blocks of the first pass against the oracle, bit for bit
    e = build_oracle_session(workload, n_tracks, 1, sample_blocks)
    e.play()
    same = all(np.array_equal(e.process()[0].view(np.uint32), rec["head"][b].view(np.uint32)) for b in range(len(rec["head"])))
    e.close()
    blocks_done = rec["blocks"] * rec["passes"]
    dev = _DEVICE_HEADS.get((workload, n_tracks, 0.0))
    # the device's verified step against the reference's own Engine::process, directly (bit-exact when the render added in
    # the reference's order — verify.summation says which)
    dev_eq = None if dev is None else bool(np.array_equal(dev[:len(rec["head"])].view(np.uint32), rec["head"].view(np.uint32)))
    dev_rms = None if dev is None else float(np.sqrt(np.mean((dev[:len(rec["head"])].astype(np.float64) - rec["head"]) ** 2)))
    return {"value": blocks_done * F / rec["seconds"], "unit": "frames/s", "cores": 1, "kind": "reference",
            "sample": f"{n_tracks} tracks x {rec['blocks']} blocks x {rec['passes']} passes ({workload}), {rec['seconds']:.1f} s of "
                      f"CPU work in the reference's own Engine::process (oracle/_ref/wbref_engine, g++ -O2), its one audio thread",
            "us_per_block": 1e6 * rec["seconds"] / blocks_done,
            "head_blocks_equal_oracle": bool(same),
            "device_head_blocks_equal_reference": dev_eq, "device_head_rms_vs_reference": dev_rms,
            "ref_driver_recipe_sha16": want, "ref_binary_sha16": exe_sha}


def cpu_all_cores(O, L, workload, n_tracks, keep, sample_blocks, src_rate, n_buses, fmt, seed, amp, passes=64):
    """'Best CPU' beside the reference-shaped single-thread number: the same oracle with the tracks
    split over one thread per physical core, each thread mixing its own shard (NOT the reference's threading —
    its engine is single-threaded by design; the final cross-shard sum of 2x512 floats is not timed)."""
    import threading
    _, _, tracks = track_layout(workload, n_tracks, 0, 1, sample_blocks)
    P = max(1, min((os.cpu_count() or 2) // 2, n_tracks // 16))
    beat_frames = SR * 60.0 / 120.0
    engines = []
    for p in range(P):
        t0, t1 = p * n_tracks // P, (p + 1) * n_tracks // P
        e = O.OracleEngine(2, F, SR)
        e.set_bpm(120.0)
        for i, t in enumerate(range(t0, t1)):
            (gt, tfmt, trate, v, pan, bus, clips) = tracks[t]
            frames = len(keep[t][0]) - 16
            sid = e.add_sample(tfmt, 2, trate, frames, keep[t])
            e.add_track()
            e.set_volume(i, v)
            e.set_pan(i, pan)
            e.add_audio_clip(i, 0.0, (sample_blocks + 1) * F / beat_frames, 0.0, sid, 1.0, 1.0)
        engines.append(e)
    start = threading.Barrier(P + 1)

    def work(e):
        out = [np.zeros(F, np.float32) for _ in range(2)]
        ptrs = O.planar_ptrs(out)
        start.wait()
        for _ in range(passes):
            e.play()
            for _ in range(sample_blocks):
                L.wbo_engine_process(e.e, ptrs, None)
            e.stop()

    threads = [threading.Thread(target=work, args=(e,)) for e in engines]
    for th in threads:
        th.start()
    start.wait()
    t0 = time.perf_counter()
    for th in threads:
        th.join()
    dt = time.perf_counter() - t0
    for e in engines:
        e.close()
    return {"value": passes * sample_blocks * F / dt, "unit": "frames/s", "cores": P, "kind": "port",
            "sample": f"{n_tracks} tracks split over {P} threads x {sample_blocks} blocks x {passes} passes, {dt:.2f} s wall",
            "note": "not the reference's threading (its engine is single-threaded); sub-bus order ignored"}


_DEVICE_HEADS = {}   # (workload, tracks, clip_blocks) -> the device's master of blocks 0-3 of the verified step


def verify_blocks_of(K, tail):
    """which blocks of a K-block step are compared with the oracle: the first 8; with `tail` also one per 256 and the last 8"""
    head = list(range(min(8, K)))
    if not tail or K <= 16:
        return head
    return sorted(set(head + list(range(255, K, 256)) + list(range(K - 8, K))))


def verify_render(eng, host_master, workload, n_tracks, rank, world, K, clip_blocks, session_blocks, chain=False, tail=True):
    """What did the timed loop compute?  After it, the transport is rewound (Engine::stop + play), one more step of K
    blocks is rendered exactly like the timed ones, and blocks of it — the first 8 and, `tail`, one per 256 and the LAST 8
    (chain words, epoch tags and the XCD hand-over of a chained render all act far behind its head) — are compared with the
    CPU oracle built from the same seeds: the device sequencer's stream-call log and the per-track peaks bit for bit, the
    master bit for bit when the render adds in the reference's order, within 1e-6 RMS otherwise.  Outside the timed
    region; the oracle only checks (bench.py's timed path never touches it).  tests/oracle_tail.py: the oracle's tracks
    sharded over the host's cores between compared blocks, one sequential Engine::process on them."""
    import oracle_tail as OT
    from whitebox_amd.engine import plan_rows_of_blocks
    t_start = time.perf_counter()
    check = verify_blocks_of(K, tail)
    res = {"blocks": len(check), "tracks": n_tracks * world, "head_blocks": min(8, K),
           "tail_blocks": [b for b in check if b >= 8], "last_block_checked": check[-1]}
    L = eng.L
    ng, longest, ref = C.c_uint32(), C.c_uint32(), C.c_int()
    L.wbx_render_order(eng.ctx.h, K, C.byref(ng), C.byref(longest), C.byref(ref))
    res["summation"] = (f"{ng.value} workgroup-level group(s) per block, longest {longest.value} tracks: "
                        + ("the reference's order" if ref.value else "grouped order"))
    _, _, n_buses, _ = WORKLOADS[workload]
    descs = []
    for r in range(world):
        seed, amp, tracks = track_layout(workload, n_tracks, r, world, session_blocks, clip_blocks)
        for (gt, tfmt, trate, v, p, bus, clips) in tracks:
            frames = int(math.ceil((session_blocks + 2) * F * (WORKLOADS[workload][1] / SR))) + 64   # build_device_session's
            descs.append(OT.TrackDesc(seed, gt, tfmt, 2, trate, frames, amp, v, p, False, (r * n_buses + bus) if n_buses else -1,
                                      [(a, b, off, 1.0, 1.0) for (a, b, off) in clips]))
    want = OT.oracle_at_blocks(descs, check, block=F, channels=2, sample_rate=SR, bpm=120.0, n_buses=n_buses * world)
    got = host_master.array[:K * 2 * F].reshape(K, 2, F)[check].copy()
    if world == 1 and check[:4] == [0, 1, 2, 3] and ref.value:      # (the first such step: the headline's; renders in the reference's order)
        _DEVICE_HEADS.setdefault((workload, n_tracks, float(clip_blocks or 0.0)), got[:4].copy())   # cpu_reference() compares the reference's own with it
    om = np.stack([want[b][0] for b in check])
    d = got.astype(np.float64) - om.astype(np.float64)
    res["rms"] = float(np.sqrt(np.mean(d * d)))
    res["max_abs"] = float(np.abs(d).max())
    res["master_bit_exact"] = bool(np.array_equal(got.view(np.uint32), om.view(np.uint32)))
    # peaks and plan rows: this rank's tracks (the other ranks' never leave their GPU)
    first = rank * n_tracks
    _, pk, _ = eng.ctx.fetch(peaks=True)
    res["peaks_equal"] = all(bool(np.array_equal(pk[b], want[b][1][first:first + n_tracks])) for b in check)
    rows = plan_rows_of_blocks(eng.fetch_plan_array(), check)
    n_rows, rows_ok = 0, True
    for b in check:
        mine = [(t - first, *rest) for (t, *rest) in want[b][2] if first <= t < first + n_tracks]
        n_rows += len(mine)
        rows_ok = rows_ok and [tuple(x) for x in rows[b]] == [tuple(x) for x in mine]
    res["plan_rows_equal"] = bool(rows_ok)
    res["plan_rows"] = n_rows
    res["ok"] = bool(res["peaks_equal"] and res["plan_rows_equal"]
                     and (res["master_bit_exact"] if (ref.value and (world == 1 or chain)) else res["rms"] <= 1e-6))
    res["seconds"] = time.perf_counter() - t_start
    return res


def rccl_log_path(rank):
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"wbx_rccl_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}_r{rank}.log")


def parse_rccl_transports(path, rank):
    """What carried this rank's exchange, from RCCL's own log (NCCL_DEBUG=INFO): every connection it sets up is announced as
    `<a>[dev] -> <b>[dev] ... via P2P/IPC` (peer-to-peer: xGMI / PCIe between two devices), `via SHM/...` (host shared memory)
    or `via NET/Socket/n` / `via NET/IB/n` (a network transport: what ranks that share one device fall back to).
    -> {"kinds": ["p2p" | "shm" | "socket" | "net:<name>", ...], "peers": {peer: [kind, ...]}, "lines": n}"""
    import re
    kinds, peers, n = set(), {}, 0
    try:
        text = open(path, errors="replace").read()
    except OSError:
        return {"kinds": [], "peers": {}, "lines": 0, "note": "no RCCL log"}
    for ln in text.splitlines():
        m = re.search(r"(\d+)\[[^\]]*\] -> (\d+)\[[^\]]*\].*? via (\S+)", ln)
        if not m:
            continue
        n += 1
        a, b, via = int(m.group(1)), int(m.group(2)), m.group(3)
        v = via.upper()
        kind = ("p2p" if v.startswith("P2P") else "shm" if v.startswith("SHM") else "socket" if v.startswith("NET/SOCKET")
                else "net:" + via.split("/")[1] if v.startswith("NET/") and "/" in via else via.lower())
        kinds.add(kind)
        peer = b if a == rank else a
        peers.setdefault(str(peer), set()).add(kind)
    return {"kinds": sorted(kinds), "peers": {k: sorted(v) for k, v in sorted(peers.items())}, "lines": n}


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n, rank_cmd=None, poll_s=0.05, grace_s=5.0):
    """`python bench.py --gpus N` run plainly: start the N ranks (one process per GPU) and pass rank 0's line through.
    A rank that fails ends the launch: the others (which would sit in the RCCL rendezvous) are terminated — killed if
    they ignore that for `grace_s` seconds — the rendezvous file is removed, and the first failing exit code is returned.
    `rank_cmd`: the command of one rank (tests pass a stub; default: this script with the same arguments)."""
    import subprocess
    if rank_cmd is None:
        import whitebox_amd as W
        have = W.lib().wbx_device_count()
        # (WBX_SHARE_DEVICE=1, an experiment aid: the ranks share the devices there are — RCCL decides whether it accepts that)
        if have < n and os.environ.get("WBX_SHARE_DEVICE") != "1":
            raise SystemExit(f"bench.py --gpus {n} needs {n} gfx950 devices on this node, {have} visible "
                             "(one process per GPU; --force-dist-path runs the multi-GPU code path on one)")
        rank_cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    port = free_port()
    rdzv = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"wbx_rdzv_{port}_{os.getpid()}")
    nonce = f"{os.getpid()}:{port}:{time.time_ns()}"        # names this launch: a stale file of an earlier one is ignored
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), WBX_RDZV=rdzv, WBX_RDZV_NONCE=nonce)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs between processes here
        if os.environ.get("WBX_SHARE_DEVICE") == "1":
            # experiment aid (one-GPU box): every rank claims a host of its own, so that RCCL accepts ranks that share a device
            # and carries the exchange over its socket transport on the loopback interface — slow, but it is wbx_dist.hip
            # running with world > 1 between processes
            env.update(NCCL_HOSTID=f"wbx-rank-{r}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1")
        procs.append(subprocess.Popen(list(rank_cmd), env=env))
    rc = 0
    alive = list(procs)
    deadline = None
    while alive:
        time.sleep(poll_s)
        for p in list(alive):
            r = p.poll()
            if r is None:
                continue
            alive.remove(p)
            if r != 0 and rc == 0:
                rc = r
                print(f"bench.py: rank {procs.index(p)} exited with {r}: ending the other ranks", file=sys.stderr, flush=True)
                for q in alive:
                    q.terminate()        # (the processes this function started, by handle)
                deadline = time.time() + grace_s
        if deadline is not None and alive and time.time() > deadline:
            for q in alive:
                q.kill()
            deadline = None
    for path in (rdzv, ):
        try:
            os.remove(path)
        except OSError:
            pass
    return rc


def run_workload(W, synth, args, workload, rank, world, *, n_tracks, K, steps, warmup, ramp, clip_blocks=0.0,
                 use_dist=False, dist_mode=0, mem_budget=96e9, latency_blocks=0, verify=True, verify_tail=True):
    """Build the session in HBM, run warmup + ramp untimed steps straight into `steps` timed ones, return the measurements."""
    from whitebox_amd.dist import Dist, PinnedBuffer
    desc, src_rate, n_buses, fmt = WORKLOADS[workload]
    total_blocks = (warmup + ramp + steps) * K
    bytes_per_block_src = n_tracks * 2 * FMT_BYTES[fmt] * F * src_rate / SR
    session_blocks = int(min(total_blocks, max(2 * K, mem_budget // bytes_per_block_src)))
    if args.session_blocks:
        session_blocks = max(K, min(session_blocks, args.session_blocks))
    session_blocks = (session_blocks // K) * K

    t_setup = time.perf_counter()
    eng, seed, amp = build_device_session(W, synth, workload, n_tracks, K, session_blocks, rank, world, args.group_size,
                                          clip_blocks)
    host_master = PinnedBuffer(K * 2 * F)             # the final (clamped) master lands here every step
    dist = None
    if use_dist:
        # wbx_dist_*: partial masters in a ring of three device buffers, RCCL reduce (or gather + ordered add) on its own
        # high-priority stream, the root's clamp straight into pinned host memory
        # (RCCL prints a version banner on stdout when a communicator is created: keep stdout for the one JSON line)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist = Dist(eng.ctx, rank, world, dist_mode)
        finally:
            C.CDLL(None).fflush(None)      # RCCL writes through C stdio: empty its buffer while fd 1 still points at stderr
            os.dup2(saved, 1)
            os.close(saved)
    else:
        # single GPU: the sum kernel stores the clamped master straight into pinned host memory
        eng.ctx.set_master_target(host_master.ptr)
    if os.environ.get("WBX_BENCH_VERBOSE"):
        print(f"[bench] {workload}: session built in {time.perf_counter() - t_setup:.2f} s", file=sys.stderr, flush=True)

    L = W.lib()
    state = {"done": 0}

    result_rank = dist.result_rank if dist is not None else 0   # who holds the summed master (chain mode: the last rank)

    def step():
        if state["done"] + K > session_blocks:     # end of the resident session: rewind (Engine::stop + play)
            eng.stop()
            eng.play()
            state["done"] = 0
        eng.render(K)
        if dist is not None:
            dist.exchange(host_master.ptr if rank == result_rank else None)   # asynchronous, beside the next renders
        # keep the submitting thread at most 12 steps ahead of the device: far deeper, the HIP runtime stalls a
        # launch until its queue has drained (tens of ms) and the device then idles
        L.wbx_pace(eng.ctx.h, 12)
        state["done"] += K

    def drain():
        if dist is not None:
            dist.sync()
        else:
            eng.ctx.sync()

    # the submitting thread must not stall inside the timed region: a cyclic-GC pass over the session's Python
    # objects (thousands of tracks / clips) costs milliseconds — several steps' worth of GPU time.  Done BEFORE the
    # warm-up so that no idle gap separates the warm-up from the timed steps (the clocks would drop again).
    gc.collect()
    gc.freeze()
    gc.disable()
    eng.play()
    for _ in range(warmup + ramp):
        step()
    drain()
    pre_ms, pre_n = eng.ctx.kernel_time(reset=True)     # warm-up + ramp launches (for the rocprofv3 cross-check)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    enq_max = 0.0
    for _ in range(steps):
        t1 = time.perf_counter()
        step()
        enq_max = max(enq_max, time.perf_counter() - t1)
    drain()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    gc.unfreeze()
    tail_ms = eng.ctx.tail_time()
    gap_ms, gap_n = eng.ctx.gap_time()
    mix_ms, mix_n = eng.ctx.kernel_time()
    kernel_name = eng.ctx.kernel_name()                 # the instance the library launched (as rocprofv3 names it)
    seq = eng.sequencer_stats()                        # how the sequencer planned: by (track, segment) lanes? seams that missed?
    if dist is not None:
        dt = dist.max(dt)                              # MAX over ranks

    master_peak = float(np.abs(host_master.array).max()) if rank == result_rank else 0.0
    # what did the timed loop compute?  rewind, one more step like the timed ones, its head against the CPU oracle
    ver = None
    if verify:
        eng.stop()
        eng.play()
        state["done"] = 0
        step()
        drain()
        if rank == result_rank:
            ver = verify_render(eng, host_master, workload, n_tracks, rank, world, K, clip_blocks, session_blocks,
                                chain=dist is not None and dist_mode == 2, tail=verify_tail)
    exch = None
    if dist is not None:
        exch = dist.info()
        # every rank's own facts, in rank order on every rank: its device, its mix kernel and exchange times, and what RCCL says
        # carried its connections — a record of the run alone then shows N devices and their transport
        mine = {"rank": rank, "pci": eng.ctx.device_info()["pci"], "mix_ms_avg": round(mix_ms, 4), "mix_launches": int(mix_n),
                "exchange_ms_avg": round(exch["exchange_ms_avg"], 4), "transport": parse_rccl_transports(rccl_log_path(rank), rank)}
        exch["ranks"] = [json.loads(x.decode()) for x in dist.allgather(json.dumps(mine).encode(), 1024)]
        if world > 1:   # what the result rank found travels to rank 0, which prints the line
            if ver is not None:
                ver = dict(ver, tail_blocks=ver["tail_blocks"][-12:])
            got = dist.allgather(json.dumps({"verify": ver, "master_peak": master_peak}).encode(), 3072)
            if rank == 0:
                back = json.loads(got[result_rank].decode())
                ver, master_peak = back["verify"], back["master_peak"]
        dist.shutdown()
        if not os.environ.get("WBX_KEEP_RCCL_LOG"):
            try:
                os.remove(rccl_log_path(rank))
            except OSError:
                pass
    dev = eng.ctx.device_info()
    ng, longest, ref_order = eng.ctx.render_order(K)
    summation = (f"{ng} workgroup-level group(s) per block, longest {longest} tracks: "
                 + ("the reference's sequential order" if ref_order else "grouped order (within 1e-6 RMS)"))
    eng.close()
    host_master.close()

    # K = 1 latency mode, the drop-in's real operating point: the audio callback, Engine::process one block at a time,
    # on an engine configured the way a callback host configures it (max_blocks = 1: many small track groups, wbx_runtime.hip build_routing)
    lat = None
    lat_small = {}
    lat_median = {}
    lat_kernel = [""]

    def callback_latency(tracks):
        eng, _, _ = build_device_session(W, synth, workload, tracks, 1, latency_blocks + 16, rank, world, args.group_size,
                                         clip_blocks)
        out = W.AudioBuffer(F, 2)
        eng.play()
        for _ in range(8):
            eng.process(None, out, float(SR))
        # the C entry point itself, as a C++ host calls it (wbx::Engine::process is one call of it): the Python
        # wrapper's argument checks would otherwise be a tenth of the measured time
        process, handle, ptrs = W.lib().wbx_engine_process, eng.h, out._ptrs()
        # (eight runs of latency_blocks / 8 calls: the mean over all calls is the number; the median of the eight run means says
        #  whether an outlier — a clock or power-state change of the box in mid-run — is in it)
        runs = []
        per = max(1, latency_blocks // 8)
        for _ in range(8):
            t1 = time.perf_counter()
            for _ in range(per):
                if process(handle, ptrs) != 0:
                    raise RuntimeError("wbx_engine_process failed")
            runs.append((time.perf_counter() - t1) / per)
        lat_kernel[0] = eng.ctx.kernel_name()
        eng.close()
        lat_median[tracks] = sorted(runs)[len(runs) // 2]
        return sum(runs) / len(runs)

    if rank == 0 and latency_blocks > 0 and dist is None:
        lat = callback_latency(n_tracks)
        # ... and for sessions of the size a DAW project usually has (up to 64 tracks: one group, or one track per group — the
        # reference's order bit for bit either way)
        for small in (8, 64):
            if small < n_tracks:
                lat_small[str(small)] = 1e3 * callback_latency(small)

    alg = algorithmic_bytes_per_block(n_tracks, src_rate, fmt=fmt) * K
    achieved = alg / (mix_ms * 1e-3) / 1e9 if mix_ms > 0 else 0.0
    return {"dt": dt, "steps": steps, "K": K, "n_tracks": n_tracks, "mix_ms": mix_ms, "mix_n": mix_n, "pre_ms": pre_ms,
            "pre_n": pre_n, "tail_ms": tail_ms, "gap_ms": gap_ms, "gap_n": gap_n, "enq_max": enq_max, "lat": lat, "lat_small": lat_small, "lat_median": lat_median, "lat_kernel": lat_kernel[0], "alg": alg, "achieved": achieved,
            "desc": desc, "kernel_name": kernel_name, "src_rate": src_rate, "n_buses": n_buses, "fmt": fmt, "master_peak": master_peak,
            "clip_blocks": clip_blocks, "workload": workload, "verify": ver, "device": dev, "exchange": exch, "summation": summation,
            "session_blocks": session_blocks,
            "sequencer": {"renders_by_segments": seq[0], "tracks_with_a_missed_seam": seq[1], "segments_replanned": seq[2],
                          "segments_per_track": seq[3]}}


def roofline_of(r, traffic_table):
    key = f"{r['workload']}_K{r['K']}_N{r['n_tracks']}" + (f"_L{r['clip_blocks']}" if r["clip_blocks"] else "")
    traffic = traffic_table.get(key, {}).get("hbm_bytes_per_launch") if F == 512 else None   # (the PMC passes are of 512-frame blocks)
    step_ms = 1e3 * r["dt"] / r["steps"]
    return {"bound": "hbm", "achieved": r["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": r["achieved"] / HBM_PEAK_GBS,
            # the same bytes over the whole step (sequencer, launch gaps, sum and master write-out included): what the
            # job sustains end to end, next to what the dominant kernel reaches while it runs
            "frac_step": r["alg"] / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms_per_step": step_ms,
            "traffic": traffic,
            # NOT this run's counters: the PMC passes of the same command, committed under profiles/ (tools/pmc_run.sh)
            "traffic_source": "profiles/pmc_traffic.json (committed rocprofv3 --pmc passes of this command)" if traffic else None,
            "kernel": r["kernel_name"],
            "kernel_ms_avg": r["mix_ms"], "kernel_launches": int(r["mix_n"]), "sum_tail_ms_avg": r["tail_ms"],
            "mix_gap_ms_avg": r["gap_ms"], "mix_gaps_timed": int(r["gap_n"]),
            # mean over EVERY launch of the run incl. warm-up and ramp: what `rocprofv3 --stats` averages
            "kernel_ms_avg_all_launches": (r["pre_ms"] * r["pre_n"] + r["mix_ms"] * r["mix_n"]) / max(1, r["pre_n"] + r["mix_n"]),
            "kernel_launches_all": int(r["pre_n"] + r["mix_n"]),
            "algorithmic_bytes_per_launch": r["alg"]}


def config_entry(name, wl, sr, K, traffic_table, extra=""):
    d2 = WORKLOADS[wl]
    ent = {"name": name, "workload": f"{wl} — {d2[0]}" + extra,
           # like the headline's `value`: master frames/s scaled to the metric's 4096 tracks (c2 has 256)
           "value": (sr["n_tracks"] / 4096.0) * sr["steps"] * K * F / sr["dt"], "unit": "frames/s",
           "master_frames_per_s": sr["steps"] * K * F / sr["dt"], "steps": sr["steps"], "blocks_per_step": K,
           "tracks": sr["n_tracks"], "ms_per_step": 1e3 * sr["dt"] / sr["steps"], "seconds": sr["dt"],
           "roofline": roofline_of(sr, traffic_table)}
    if sr.get("verify"):
        ent["verify"] = sr["verify"]
    return ent


def compact_entry(ent):
    """what the one stdout line keeps of a `configs` entry (the full entry goes to stderr as its own JSON line): the line is
    read by tools that keep only its tail"""
    rf, v = ent["roofline"], ent.get("verify")
    out = {"workload": ent["name"],
           "value": ent["value"], "ms_per_step": ent["ms_per_step"], "blocks_per_step": ent["blocks_per_step"], "tracks": ent["tracks"],
           "roofline": {"frac": rf["frac"], "frac_step": rf["frac_step"], "achieved": rf["achieved"], "traffic": rf["traffic"],
                        "kernel": rf["kernel"], "kernel_ms_avg": rf["kernel_ms_avg"]}}
    if v:
        out["verify"] = {"ok": v["ok"], "master_bit_exact": v["master_bit_exact"], "rms": v["rms"], "blocks": v["blocks"],
                         "last_block_checked": v["last_block_checked"], "peaks_equal": v["peaks_equal"],
                         "plan_rows_equal": v["plan_rows_equal"]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ramp-steps", type=int, default=None, help="further untimed steps straight before the timed ones "
                    "(default: 10240 blocks' worth, at least 6): the device needs ~20 ms of continuous load to reach "
                    "its sustained clocks, and drops them again in any idle gap")
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--tracks", type=int, default=None, help="tracks per GPU (default 4096; 256 for c2)")
    ap.add_argument("--blocks", type=int, default=2048, help="512-frame blocks per step (one device pass).  Renders of "
                    ">= 1024 blocks add every block's tracks in the reference's strictly sequential order (bit-exact "
                    "master: 128-track workgroups, each continuing the running sum of the one before it — at full rate "
                    "from about 1536 blocks on); shorter ones in groups of 128 (within 1e-6 RMS)")
    ap.add_argument("--group-size", type=int, default=0)
    ap.add_argument("--clip-blocks", type=float, default=0.0, help="side measurement: cut every track into back-to-back "
                    "clips of this many blocks (0: one clip per track, the BASELINE.json configs)")
    ap.add_argument("--block-frames", type=int, default=512, help="frames per block (BASELINE.json: 512; other sizes "
                    "are side measurements)")
    ap.add_argument("--session-blocks", type=int, default=0, help="length of the resident session in blocks "
                    "(0 = as long as the run needs, capped by memory); the transport rewinds at its end")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--latency-blocks", type=int, default=800, help="K=1 Engine::process calls timed after the run (per session size)")
    ap.add_argument("--force-dist-path", action="store_true",
                    help="run the multi-GPU code path (RCCL exchange + clamp on root) even with one rank")
    ap.add_argument("--dist-mode", default="auto", choices=["auto", "reduce", "ordered", "chain"],
                    help="exchange: a chain — rank g continues rank g-1's running master (the reference's sequential order "
                    "across GPUs: bit-exact; `auto` takes it for renders of >= 1024 blocks, whose shards add in that order too); "
                    "one ncclReduce (`auto` for shorter renders); or gather + fixed-order add on the root (bit-reproducible).  "
                    "reduce / ordered add SHARD sums: within 1e-6 RMS up to about half of full scale (profiles/r03_level_probe.txt)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: BASELINE configs[4]'s 32768 tracks split over "
                    "the N ranks (default: weak scaling, 4096 tracks per GPU)")
    ap.add_argument("--no-configs", action="store_true", help="skip the short runs of the other single-GPU configurations "
                    "(c2, c4, c3 cut into clips, i16r, the sustained run) that fill the line's `configs` object")
    ap.add_argument("--no-verify", action="store_true", help="skip the check of the rendered step against the CPU oracle "
                    "(after the timed loop, outside the timed region)")
    ap.add_argument("--no-verify-tail", action="store_true", help="compare only the first 8 blocks of the verified step with the "
                    "oracle (default: also one block per 256 and the last 8 — the oracle's tracks run sharded over the host's "
                    "cores between compared blocks, a few seconds per configuration)")
    ap.add_argument("--sustain-blocks", type=int, default=655360, help="blocks of the sustained-clock run in `configs` "
                    "(2 000 steps of 256 blocks' worth: >= 1.5 s of uninterrupted load)")
    args = ap.parse_args()
    global F
    F = args.block_frames

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} inside a launch of {world} ranks")
    if world > 1 and os.environ.get("WBX_SHARE_DEVICE") == "1":   # (also under another launcher, e.g. torch.distributed.run)
        os.environ.setdefault("NCCL_HOSTID", f"wbx-rank-{rank}")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")

    import whitebox_amd as W
    from whitebox_amd import synth
    if W.lib().wbx_device_count() <= rank_device(rank):
        raise SystemExit(f"rank {rank}: no gfx950 device {rank_device(rank)} ({W.lib().wbx_device_count()} visible) - "
                         "the mix path has no CPU implementation")

    n_tracks = args.tracks or (256 if args.workload == "c2" else 4096)
    if args.strong:
        if 32768 % world:
            raise SystemExit("--strong: 32768 tracks do not split evenly over this many ranks")
        n_tracks = 32768 // world
    K = args.blocks
    ramp = args.ramp_steps if args.ramp_steps is not None else max(6, 10240 // K)
    use_dist = world > 1 or args.force_dist_path
    if args.dist_mode == "auto":
        args.dist_mode = "chain" if K >= 1024 else "reduce"
    dist_mode = {"reduce": 0, "ordered": 1, "chain": 2}[args.dist_mode]
    if use_dist:   # RCCL's own account of its connections (parse_rccl_transports), one file per rank
        os.environ["NCCL_DEBUG"] = "INFO"          # (whatever level the environment asked for: the connection lines are INFO)
        os.environ["NCCL_DEBUG_SUBSYS"] = "INIT,P2P,SHM,NET"
        os.environ["NCCL_DEBUG_FILE"] = rccl_log_path(rank)
    r = run_workload(W, synth, args, args.workload, rank, world, n_tracks=n_tracks, K=K, steps=args.steps,
                     warmup=args.warmup, ramp=ramp, clip_blocks=args.clip_blocks, use_dist=use_dist,
                     dist_mode=dist_mode, latency_blocks=args.latency_blocks, verify=not args.no_verify, verify_tail=not args.no_verify_tail)
    # every rank says what it ran on (stderr: stdout carries rank 0's one JSON line)
    print(json.dumps({"rank": rank, "world": world, "device": r["device"], "tracks": n_tracks, "exchange": r["exchange"],
                      "mix_ms_avg": r["mix_ms"], "seconds": r["dt"]}), file=sys.stderr, flush=True)
    if rank != 0:
        return

    dt = r["dt"]
    master_frames = args.steps * K * F
    total_tracks = n_tracks * world
    value = (total_tracks / 4096.0) * master_frames / dt
    traffic_table = {}
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic_table = json.load(open(tpath))
        except Exception:
            traffic_table = {}
    desc, src_rate, n_buses, fmt = WORKLOADS[args.workload]
    ver = r["verify"]
    line = {
        "metric": "stereo fp32 frames/sec mixed (4096 tracks @ 512-frame blocks)",
        "value": value, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ramp_steps": ramp,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
        "dtype": "f32" if fmt == "f32" else f"f32 (clips stored as {fmt})", "data": "synthetic",
        "config": {"workload": f"{args.workload} — {desc}", "tracks_per_gpu": n_tracks, "total_tracks": total_tracks,
                   "blocks_per_step": K, "block_frames": F, "dst_rate": SR, "src_rate": src_rate, "clip_format": fmt,
                   "sub_buses": n_buses, "group_size": args.group_size or "library default",
                   "summation": r["summation"],
                   "clip_blocks": args.clip_blocks or None,
                   "session_level": f"amp=0.25/sqrt({total_tracks})", "parallelism": f"tracks sharded x{world}",
                   "exchange": ((f"libwbx wbx_dist_exchange over RCCL, mode {args.dist_mode}: "
                                 + ("rank g continues rank g-1's running master (ncclSend / ncclRecv), the last rank clamps — the "
                                    "reference's order across GPUs, bit-exact at any level" if args.dist_mode == "chain" else
                                    "SHARD sums are added (not the reference's association): within 1e-6 RMS up to about half of "
                                    "full scale, 1.5e-6 at amp = 1/sqrt(N) over 32768 tracks (profiles/r03_level_probe.txt); "
                                    "--dist-mode chain is bit-exact")) if use_dist else None)},
        "master_frames_per_s": master_frames / dt,
        "track_frames_per_s": total_tracks * master_frames / dt,
        "realtime_factor": master_frames / dt / SR,
        "host_enqueue_ms_max": 1e3 * r["enq_max"],
        "master_peak": r["master_peak"],
        "sequencer": r["sequencer"],
        "roofline": roofline_of(r, traffic_table),
    }
    if ver is not None:
        line["verify"] = ver
    if use_dist and r["exchange"]:
        # what the exchange saw: RCCL's world size, which device every rank ran on, the exchange's own time
        ranks = r["exchange"].get("ranks", [])
        kinds = sorted({k for x in ranks for k in x["transport"]["kinds"]})
        line.update({"rccl_world": r["exchange"]["world"], "devices": r["exchange"]["devices"], "tracks_per_gpu": n_tracks,
                     "exchange_ms_avg": r["exchange"]["exchange_ms_avg"],
                     # N ranks on N devices?  and what carried the exchange, per RCCL's own log ("p2p" = device to device: xGMI /
                     # PCIe peer access; "socket" = the loopback network transport ranks that SHARE a device fall back to)
                     "distinct_devices": len(set(r["exchange"]["devices"])) == r["exchange"]["world"],
                     "transport": (kinds[0] if len(kinds) == 1 else "none (one rank)" if world == 1 and not kinds
                                   else "+".join(kinds) if kinds else "unknown"),
                     "ranks": ranks})
    if r["lat"] is not None:
        line["latency_mode"] = {"blocks_per_call": 1, "ms_per_block": 1e3 * r["lat"], "frames_per_s": F / r["lat"],
                                "calls": args.latency_blocks, "launches_per_call": 1 if "callback_kernel" in (r.get("lat_kernel") or "") else 3,
                                "ms_per_block_median_of_8_runs": {str(k): 1e3 * v for k, v in r["lat_median"].items()},
                                # the same call for sessions of 8 / 64 tracks (8: one group, the mix workgroup stores the master itself; 64: one track per group)
                                "ms_per_block_small_sessions": r.get("lat_small") or None}

    # the other single-GPU configurations of BASELINE.json (configs[1], configs[3]), the headline session cut into
    # clips, the 16-bit resampled session, the round-2 operating point (256-block renders, grouped order) and a
    # sustained run: each with its own roofline from its own HIP-event kernel times, each checked against the oracle
    failed = [] if (ver is None or ver["ok"]) else ["headline"]
    if world == 1 and not use_dist and not args.no_configs and args.workload == "c3" and not args.clip_blocks and F == 512:
        subs = {}
        for name, wl, kw in (("c2", "c2", dict(n_tracks=256)), ("c4", "c4", dict(n_tracks=4096)),
                             ("c3_clips5.3", "c3", dict(n_tracks=4096, clip_blocks=5.3)),
                             ("i16r", "i16r", dict(n_tracks=4096))):
            # (c2: a step is 0.4 ms — 200 of them, or the one drain of the last master's copy-out at the end of the timed
            #  region is 4 % of what is measured)
            sr = run_workload(W, synth, args, wl, 0, 1, K=K, steps=200 if wl == "c2" else 20, warmup=3, ramp=ramp * (3 if wl == "c2" else 1),
                              mem_budget=40e9, verify=not args.no_verify, verify_tail=not args.no_verify_tail, **kw)
            subs[name] = config_entry(name, wl, sr, K, traffic_table,
                                      f", every track cut into clips of {kw['clip_blocks']} blocks" if kw.get("clip_blocks") else "")
        # renders of 256 blocks: the grouped summation order (within 1e-6 RMS), the operating point of rounds 1-2
        sr = run_workload(W, synth, args, "c3", 0, 1, n_tracks=4096, K=256, steps=20, warmup=3, ramp=40, mem_budget=40e9,
                          verify=not args.no_verify, verify_tail=not args.no_verify_tail)
        subs["c3_K256_grouped"] = config_entry("c3_K256_grouped", "c3", sr, 256, traffic_table, ", 256-block renders (128-track groups)")
        # sustained clocks: the headline configuration for >= 1.5 s of consecutive steps
        s_steps = max(20, -(-args.sustain_blocks // K))
        sr = run_workload(W, synth, args, "c3", 0, 1, n_tracks=4096, K=K, steps=s_steps, warmup=3, ramp=ramp, verify=False)
        subs["c3_sustained"] = config_entry("c3_sustained", "c3", sr, K, traffic_table,
                                            f", {s_steps} consecutive steps = {s_steps * K} blocks")
        for name, ent in subs.items():
            if ent.get("verify") and not ent["verify"]["ok"]:
                failed.append(name)
        # the full entries: their own JSON line on stderr (tools/profile_round.sh keeps it); the stdout line carries the compact form
        print(json.dumps({"configs_full": subs}), file=sys.stderr, flush=True)
        line["configs"] = {k: compact_entry(v) for k, v in subs.items()}
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args.workload, n_tracks, args.cpu_seconds)
    # the headline once more at the very END of the line: a reader that keeps only the tail of a long line still sees it
    rf = line["roofline"]
    line["headline"] = {"value": value, "unit": "frames/s", "n_gpus": world, "ms_per_step": line["ms_per_step"],
                        "roofline_frac": rf["frac"], "roofline_frac_step": rf["frac_step"], "kernel_ms_avg": rf["kernel_ms_avg"],
                        "verify_ok": None if ver is None else ver["ok"],
                        "master_bit_exact": None if ver is None else ver["master_bit_exact"],
                        "last_block_checked": None if ver is None else ver["last_block_checked"],
                        "device_head_equals_reference_engine": line.get("cpu_baseline", {}).get("device_head_blocks_equal_reference"),
                        "cpu_baseline_kind": line.get("cpu_baseline", {}).get("kind"),
                        "latency_ms_per_block": line.get("latency_mode", {}).get("ms_per_block"),
                        "cpu_baseline_frames_per_s": line.get("cpu_baseline", {}).get("value"),
                        "configs_ok": [k for k in line.get("configs", {}) if line["configs"][k].get("verify", {}).get("ok")],
                        "configs_failed": failed}
    print(json.dumps(line), flush=True)
    if failed:
        print(f"bench.py: the rendered head differs from the CPU oracle in: {', '.join(failed)}", file=sys.stderr, flush=True)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
