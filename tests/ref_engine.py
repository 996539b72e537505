"""Script interface to oracle/_ref/wbref_engine — the reference's OWN clip sequencer and block driver (Track::process_event,
Track::process, Engine::process ... cut out of engine/track.cpp / engine.cpp where they lie and compiled unmodified:
oracle/Makefile, oracle/ref_engine_driver.cpp) — and the same script replayed on the oracle.  Test infrastructure only.

A script is a list of operations; `run` operations process blocks.  The reference answers every non-run operation with a
status: 1 taken, 0 refused (the edit would need Engine::reserve_track_region, which holds a spdlog line and is not in the cut),
2 bad argument.  The oracle side PREDICTS the refusals with its own restatement of add_to_cliplist's early exits and
Track::query_clip_by_range, so the prediction is itself compared."""
import os
import struct
import subprocess
import tempfile
from typing import List, Optional

import numpy as np

import oracle_ffi as O

EXE = os.path.join(O.ORACLE_DIR, "_ref", "wbref_engine")


def available(build: bool = True) -> bool:
    """is the reference executable there?  build=True (the tests, this container): oracle/Makefile's `ref` target runs first where
    /root/reference exists; build=False (bench.py): the prebuilt file or nothing — a bench run reads nothing of the reference"""
    if build and not O.build_ref():
        return False
    if not os.path.exists(EXE):
        return False
    if not os.access(EXE, os.X_OK):          # (a copy of the tree that dropped the mode bits)
        try:
            os.chmod(EXE, 0o755)
        except OSError:
            return False
    return os.access(EXE, os.X_OK)


class Script:
    """operations as tuples; samples as (fmt, channels, rate, frames, [planar arrays incl. 16 pad frames])"""

    def __init__(self, channels=2, block=512, rate=48000, bpm=120.0):
        self.channels, self.block, self.rate = channels, block, rate
        self.ops = [("cfg", channels, block, rate), ("bpm", float(bpm))]
        self.samples = []

    def add_sample(self, fmt, channels, rate, frames, data, gen=None):
        """gen = (seed, seed_track, amp) of whitebox_amd.synth's keyed generator when the data came from it (fixtures store the
        key, not the audio)"""
        self.samples.append((fmt, channels, rate, frames, data, gen))
        self.ops.append(("sample", len(self.samples) - 1))
        return len(self.samples) - 1

    def op(self, *a):
        self.ops.append(tuple(a))


def script_to_json(s: Script) -> str:
    """ops with their floats as hex strings (exact), samples as generator keys"""
    import json
    def enc(x):
        if isinstance(x, (float, np.floating)):
            return {"f": float(x).hex()}
        return int(x) if isinstance(x, (int, np.integer)) and not isinstance(x, bool) else x
    assert all(smp[5] is not None for smp in s.samples)
    return json.dumps({"channels": s.channels, "block": s.block, "rate": s.rate,
                       "samples": [[smp[0], smp[1], smp[2], smp[3], list(smp[5][:2]) + [float(smp[5][2]).hex()]] for smp in s.samples],
                       "ops": [[enc(x) for x in o] for o in s.ops]})


def script_from_json(text: str) -> Script:
    import json
    from whitebox_amd import synth
    d = json.loads(text)
    s = Script(d["channels"], d["block"], d["rate"])
    s.ops = []
    for fmt, ch, rate, frames, (seed, seed_track, amp) in d["samples"]:
        amp = float.fromhex(amp)
        spec = synth.SessionSpec(name="s", n_tracks=1, seed=seed, samples=[synth.SampleSpec(seed_track, ch, rate, frames, fmt, amp)],
                                 clips=[], volumes_db=[0.0], pans=[0.0], mutes=[False])
        s.samples.append((fmt, ch, rate, frames, spec.sample_data(0), (seed, seed_track, amp)))
    for o in d["ops"]:
        s.ops.append(tuple(float.fromhex(x["f"]) if isinstance(x, dict) else x for x in o))
    return s


def script_from_spec(spec, n_blocks: int) -> Script:
    """a whitebox_amd.synth.SessionSpec (the BASELINE configs, the fuzz generators' sessions ...) as a script: fp32 samples come
    from the driver's own copy of the keyed generator (`synth` operation: no audio in the script or the data file), other
    formats through the data file; clips in list order, as build_oracle_engine / build_engine add them"""
    s = Script(spec.channels, spec.block, spec.sample_rate, spec.bpm)
    for i, smp in enumerate(spec.samples):
        if smp.fmt == "f32":
            s.samples.append(("f32", smp.channels, smp.rate, smp.frames, spec.sample_data(i), (spec.seed, smp.seed_track, smp.amp)))
            s.ops.append(("synth", len(s.samples) - 1))
        else:
            s.add_sample(smp.fmt, smp.channels, smp.rate, smp.frames, spec.sample_data(i), gen=(spec.seed, smp.seed_track, smp.amp))
    for t in range(spec.n_tracks):
        s.op("track")
        s.op("vol", t, float(np.float32(spec.volumes_db[t])))
        s.op("pan", t, float(np.float32(spec.pans[t])))
        if spec.mutes[t]:
            s.op("mute", t, 1)
    for c in spec.clips:
        s.op("clip", c.track, float(c.min_beat), float(c.max_beat), float(c.start_offset),
             c.sample if c.sample is not None else c.track, float(c.speed), float(np.float32(c.gain)))
    if spec.playhead_start:
        s.op("seek", float(spec.playhead_start))
    s.op("play")
    s.op("run", n_blocks)
    return s


def script_from_bus_spec(spec, n_blocks: int) -> Script:
    """a session with sub-buses (extension A13; config 4) for the driver's `bus` / `runbus` operations: one reference Engine per
    bus holding that bus's tracks in track order, the buses added in order — SURVEY A13's composition of reference functions"""
    assert spec.n_buses and spec.track_bus is not None
    s = Script(spec.channels, spec.block, spec.sample_rate, spec.bpm)
    for i, smp in enumerate(spec.samples):
        assert smp.fmt == "f32"
        s.samples.append(("f32", smp.channels, smp.rate, smp.frames, None, (spec.seed, smp.seed_track, smp.amp)))
        s.ops.append(("synth", len(s.samples) - 1))
    for b in range(spec.n_buses):
        s.op("bus", b)
        members = [t for t in range(spec.n_tracks) if spec.track_bus[t] == b]
        local = {t: i for i, t in enumerate(members)}
        for t in members:
            s.op("track")
            s.op("vol", local[t], float(np.float32(spec.volumes_db[t])))
            s.op("pan", local[t], float(np.float32(spec.pans[t])))
            if spec.mutes[t]:
                s.op("mute", local[t], 1)
        for c in spec.clips:
            if c.track in local:
                s.op("clip", local[c.track], float(c.min_beat), float(c.max_beat), float(c.start_offset),
                     c.sample if c.sample is not None else c.track, float(c.speed), float(np.float32(c.gain)))
    s.op("play")
    s.op("runbus", n_blocks)
    return s


def _hx(x: float) -> str:
    return float(x).hex()


def run_reference(s: Script, timeout=60, want_raw=False):
    """-> list of records: ("op", status) | ("run", [block dicts]) | ("clips", [[clip tuples] per track])"""
    blob, offs = bytearray(), []
    synth_ops = {o[1] for o in s.ops if o[0] == "synth"}
    for i, (fmt, ch, rate, frames, data, _g) in enumerate(s.samples):
        offs.append(len(blob))
        if i in synth_ops:
            continue                      # generated by the driver from its key
        for c in range(ch):
            blob += np.ascontiguousarray(data[c][:frames]).tobytes()
    lines = []
    for o in s.ops:
        k = o[0]
        if k == "sample":
            fmt, ch, rate, frames = s.samples[o[1]][:4]
            lines.append(f"sample {O.FMT[fmt]} {ch} {rate} {frames} {offs[o[1]]}")
        elif k == "synth":
            fmt, ch, rate, frames, _d, (seed, kt, amp) = s.samples[o[1]]
            lines.append(f"synth {ch} {rate} {frames} {seed} {kt} {_hx(np.float32(amp))}")
        elif k == "clip":
            _, t, mn, mx, so, si, sp, g = o
            lines.append(f"clip {t} {_hx(mn)} {_hx(mx)} {_hx(so)} {si} {_hx(sp)} {_hx(np.float32(g))}")
        elif k == "gain":
            lines.append(f"gain {o[1]} {o[2]} {_hx(np.float32(o[3]))}")
        elif k == "move":
            lines.append(f"move {o[1]} {o[2]} {_hx(o[3])}")
        elif k == "query":
            lines.append(f"query {o[1]} {_hx(o[2])} {_hx(o[3])}")
        elif k in ("vol", "pan"):
            lines.append(f"{k} {o[1]} {float(np.float32(o[2]))!r}")
        elif k in ("bpm", "seek"):
            lines.append(f"{k} {float(o[1])!r}")
        else:
            lines.append(" ".join(str(x) for x in o))
    with tempfile.TemporaryDirectory() as d:
        sp, dp, rp = (os.path.join(d, n) for n in ("script.txt", "data.bin", "result.bin"))
        open(sp, "w").write("\n".join(lines) + "\n")
        open(dp, "wb").write(bytes(blob))
        r = subprocess.run([EXE, sp, dp, rp], timeout=timeout, capture_output=True)
        if r.returncode != 0:
            raise RuntimeError(f"wbref_engine rc={r.returncode} {r.stderr[-300:]!r}")
        raw = open(rp, "rb").read()
    return (raw, parse_results(raw, s.channels, s.block)) if want_raw else parse_results(raw, s.channels, s.block)


def parse_results(raw: bytes, C: int, F: int):
    pos, out = 0, []

    def u32():
        nonlocal pos
        v = struct.unpack_from("<I", raw, pos)[0]; pos += 4
        return v

    def f64bits():
        nonlocal pos
        v = struct.unpack_from("<Q", raw, pos)[0]; pos += 8
        return v

    while pos < len(raw):
        tag = u32()
        if tag == 0x4F500000:
            out.append(("op", u32()))
        elif tag == 0x52554E00:
            n, blocks = u32(), []
            for _ in range(n):
                assert u32() == 0x424C4B00
                b = {"block": u32()}
                b["master"] = np.frombuffer(raw, np.uint32, C * F, pos).reshape(C, F).copy(); pos += 4 * C * F
                b["playhead"], b["sample_position"] = f64bits(), f64bits()
                tr = []
                for _t in range(u32()):
                    ev = []
                    for _e in range(u32()):
                        typ, boff = u32(), u32()
                        time, speed, so = f64bits(), f64bits(), f64bits()
                        ev.append((typ, boff, time, speed, so))
                    cur = u32()
                    spd, soff = f64bits(), f64bits()
                    lv = (u32(), u32())
                    tr.append({"events": ev, "current": cur, "speed": spd, "offset": soff, "level": lv})
                b["tracks"] = tr
                blocks.append(b)
            out.append(("run", blocks))
        elif tag == 0x51525900:
            has, first, last = u32(), u32(), u32()
            fo, lo = f64bits(), f64bits()
            out.append(("query", (has, first, last) if has else (0, 0, 0)))
        elif tag == 0x52554200:
            n, nb = u32(), u32()
            blocks = []
            for _ in range(n):
                m = np.frombuffer(raw, np.uint32, C * F, pos).reshape(C, F).copy(); pos += 4 * C * F
                bus = np.frombuffer(raw, np.uint32, nb * C * F, pos).reshape(nb, C, F).copy(); pos += 4 * nb * C * F
                peak = struct.unpack_from("<f", raw, pos)[0]; pos += 4
                blocks.append({"master": m, "buses": bus, "peak": peak, "playhead": f64bits(), "sample_position": f64bits()})
            out.append(("runbus", blocks))
        elif tag == 0x42454E00:
            n, passes = u32(), u32()
            secs = struct.unpack_from("<d", raw, pos)[0]; pos += 8
            cnt = u32()
            head = np.frombuffer(raw, np.float32, cnt, pos).copy(); pos += 4 * cnt
            out.append(("bench", {"blocks": n, "passes": passes, "seconds": secs, "head": head.reshape(-1, C, F)}))
        elif tag == 0x434C5000:
            lists = []
            for _t in range(u32()):
                cl = []
                for _c in range(u32()):
                    mn, mx, so, spd = f64bits(), f64bits(), f64bits(), f64bits()
                    g, ai = u32(), u32()
                    cl.append((mn, mx, so, spd, g, ai))
                lists.append(cl)
            out.append(("clips", lists))
        else:
            raise RuntimeError(f"bad tag {tag:#x} at {pos}")
    return out


def _add_needs_trim(e: O.OracleEngine, t, mn, mx) -> bool:
    """Engine::add_to_cliplist (engine.cpp:409-461): does the add get past the three early exits AND find clips in its range?"""
    cl = e.clips(t)
    if not cl:
        return False
    if cl[-1][1] < mn:
        return False
    if cl[0][0] > mx:
        return False
    f, l = O.C.c_uint32(), O.C.c_uint32()
    return bool(e.L.wbo_track_query_clip_by_range(e.e, t, O.C.c_double(mn), O.C.c_double(mx), O.C.byref(f), O.C.byref(l)))


class Wrapped(Exception):
    """the session drives the reference into its event_length wrap (track.cpp:669: a write past the block buffer, undefined
    behaviour in the reference) — nothing to compare"""


def run_oracle(s: Script):
    """the same script on the oracle; same record shapes as run_reference (bit patterns)"""
    e = O.OracleEngine(s.channels, s.block, s.rate)
    e.enable_seglog(True)
    out, block_no = [], 0
    fb = O.f64_bits
    for o in s.ops:
        k, st = o[0], 1
        if k == "cfg":
            pass
        elif k == "bpm":
            e.set_bpm(o[1])
        elif k == "seek":
            e.set_playhead(o[1])
        elif k == "rate":
            e.e.contents.sample_rate = int(o[1])      # Engine::process is handed the new rate (engine.cpp:1576), nothing else changes
        elif k == "play":
            e.play()
        elif k == "stop":
            e.stop()
        elif k in ("sample", "synth"):
            fmt, ch, rate, frames, data = s.samples[o[1]][:5]
            e.add_sample(fmt, ch, rate, frames, data)
        elif k == "track":
            e.add_track()
        elif k == "vol":
            e.set_volume(o[1], o[2])
        elif k == "pan":
            e.set_pan(o[1], o[2])
        elif k == "mute":
            e.set_mute(o[1], o[2])
        elif k == "clip":
            _, t, mn, mx, so, si, sp, g = o
            if _add_needs_trim(e, t, mn, mx):
                st = 0
            else:
                assert e.add_audio_clip(t, mn, mx, so, si, sp, g) == 0
        elif k == "delclip":
            if o[2] >= len(e.clips(o[1])):
                st = 2
            else:
                e.delete_clip(o[1], o[2])
        elif k == "gain":
            if o[2] >= len(e.clips(o[1])):
                st = 2
            else:
                e.set_clip_gain(o[1], o[2], o[3])
        elif k == "move":
            _, t, i, rel = o
            cl = e.clips(t)
            if i >= len(cl):
                st = 2
            else:
                mn, mx = O.C.c_double(), O.C.c_double()
                e.L.wbo_calc_move_clip(O.C.c_double(cl[i][0]), O.C.c_double(cl[i][1]), O.C.c_double(rel), O.C.c_double(0.0),
                                       O.C.byref(mn), O.C.byref(mx))
                f, l = O.C.c_uint32(), O.C.c_uint32()
                if rel != 0.0 and e.L.wbo_track_query_clip_by_range(e.e, t, mn, mx, O.C.byref(f), O.C.byref(l)):
                    st = 0
                else:
                    e.move_clip(t, i, float(rel))
        elif k == "query":
            f, l = O.C.c_uint32(), O.C.c_uint32()
            has = int(bool(e.L.wbo_track_query_clip_by_range(e.e, o[1], O.C.c_double(o[2]), O.C.c_double(o[3]), O.C.byref(f), O.C.byref(l))))
            out.append(("query", (has, f.value, l.value) if has else (0, 0, 0)))
            continue
        elif k == "deltrack":
            e.delete_track(o[1])
        elif k == "movetrack":
            e.move_track(o[1], o[2])
        elif k == "solo":
            e.solo_track(o[1])
        elif k == "run":
            blocks = []
            for _ in range(o[1]):
                m, _b = e.process()
                for sg in e.seglog():
                    if sg[1] + sg[2] > s.block:
                        raise Wrapped()
                nt = e.e.contents.n_tracks
                tr = []
                for t in range(nt):
                    T = e.track(t)
                    ev = [(x.type, x.buffer_offset, fb(x.time), fb(x.speed) if x.type == 2 else 0,
                           int(x.sample_offset) if x.type == 2 else 0) for x in T.events[:T.n_events]]
                    tr.append({"events": ev, "current": T.current_event.type, "speed": fb(T.sampler.playback_speed),
                               "offset": fb(T.sampler.sample_offset), "level": (O.f32_bits(T.level[0]), O.f32_bits(T.level[1]))})
                blocks.append({"block": block_no, "master": m.view(np.uint32).copy(), "playhead": fb(e.playhead),
                               "sample_position": fb(e.sample_position), "tracks": tr})
                block_no += 1
            out.append(("run", blocks))
            continue
        elif k == "clips":
            nt = e.e.contents.n_tracks
            out.append(("clips", [[(fb(c[0]), fb(c[1]), fb(c[2]), fb(c[3]), O.f32_bits(c[4]), c[5]) for c in e.clips(t)]
                                  for t in range(nt)]))
            continue
        else:
            st = 2
        out.append(("op", st))
    e.close()
    return out


def compare(ref, orc, what="") -> Optional[str]:
    """first difference between two record lists, or None"""
    if len(ref) != len(orc):
        return f"{what}: {len(ref)} records against {len(orc)}"
    for i, (r, o) in enumerate(zip(ref, orc)):
        if r[0] != o[0]:
            return f"{what}: record {i} kinds {r[0]} / {o[0]}"
        if r[0] == "op":
            if r[1] != o[1]:
                return f"{what}: operation {i} status reference {r[1]} oracle {o[1]}"
        elif r[0] == "query":
            if r[1] != o[1]:
                return f"{what}: range query {i}: reference {r[1]} oracle {o[1]}"
        elif r[0] == "clips":
            if r[1] != o[1]:
                return f"{what}: clip lists differ at record {i}: {r[1]} / {o[1]}"
        else:
            for br, bo in zip(r[1], o[1]):
                b = br["block"]
                for key in ("playhead", "sample_position"):
                    if br[key] != bo[key]:
                        return f"{what}: block {b} {key} {br[key]:#x} / {bo[key]:#x}"
                if len(br["tracks"]) != len(bo["tracks"]):
                    return f"{what}: block {b} track count"
                for t, (tr, to) in enumerate(zip(br["tracks"], bo["tracks"])):
                    for key in ("events", "current", "speed", "offset", "level"):
                        if tr[key] != to[key]:
                            return f"{what}: block {b} track {t} {key}: reference {tr[key]} oracle {to[key]}"
                if not np.array_equal(br["master"], bo["master"]):
                    d = np.argwhere(br["master"] != bo["master"])
                    return f"{what}: block {b} master differs at {d[:4].tolist()} ({len(d)} samples)"
    return None
