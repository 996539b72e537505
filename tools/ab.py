#!/usr/bin/env python3
"""tools/ab.py — A/B and sweep runs of bench.py inside ONE gpurun call (same box, same thermal state).

    python tools/ab.py PRESET [-r REPEATS] [-o OUT.jsonl]
    python tools/ab.py run "<label>;<ENV=1 ENV2=2>;<bench args>" ...      ad-hoc cases

Every case is one `python bench.py <args> --no-cpu-baseline --no-configs` (+ `--latency-blocks 0` unless the case
measures the callback path) with the given environment; cases are interleaved REPEATS times (A B A B, not A A B B) so
that drift of the box hits both sides.  One summary line per run; `-o` appends the raw bench lines (with the label).

The presets are the experiments DESIGN-experiments.md refers to (one shell script each until round 3).
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STD = "--steps 20 --warmup 3 --ramp-steps 40"


def case(label, env="", args="", lat=False):
    return {"label": label, "env": env, "args": args, "lat": lat}


def presets(other_lib=None):
    P = {}
    # round 3: the reference's summation order at full width — one workgroup walks all N rows of its block in track order;
    # parallelism comes from the K blocks of a launch instead of from track groups
    P["exact"] = [case(f"{w} exact K={k}", "", f"--workload {w} --group-size 4096 --blocks {k} --steps {max(4, 5120 // k)} "
                       f"--warmup 2 --ramp-steps {max(4, 10240 // k)}")
                  for w in ("c3", "c4") for k in (256, 512, 768, 1024, 1536, 2048)] + \
                 [case(f"{w} groups128 K=256", "", f"--workload {w} {STD}") for w in ("c3", "c4")]
    P["exact_variants"] = [case(f"{w} exact K={k} variant={v}", f"WBX_MIX_VARIANT={v}",
                                f"--workload {w} --group-size 4096 --blocks {k} --steps {max(4, 5120 // k)} --warmup 2 "
                                f"--ramp-steps {max(4, 10240 // k)}")
                           for w in ("c3", "c4") for k in (1024, 2048) for v in (24, 43, 82, 1000)]
    # whole-list walks (the library's choice for renders of >= 1024 blocks) against 128-track groups at the same render length
    P["exact_vs_grouped"] = [case(f"{w}{' L=%s' % l if l else ''} K=1024 {'grouped' if g else 'whole lists'}",
                                  "WBX_EXACT_MIN_BLOCKS=0" if g else "",
                                  f"--workload {w} {'--clip-blocks %s' % l if l else ''} --blocks 1024 --steps 6 --warmup 2 --ramp-steps 10")
                             for (w, l) in (("c3", 0), ("c4", 0), ("c2", 0), ("u4096", 0), ("i16", 0), ("i16r", 0), ("i24r", 0),
                                            ("mixr", 0), ("mixfmt", 0), ("d96", 0), ("c3", 5.3), ("i16", 5.3))
                             for g in (0, 1)]
    # the reference's order three ways at several render lengths: chained pieces (default), whole-list walks (WBX_CHAIN=0),
    # and the grouped order beside them (WBX_EXACT_MIN_BLOCKS=0)
    P["chain"] = [case(f"{w}{' L=%s' % l if l else ''} K={k} {lab}", env,
                       f"--workload {w} {'--clip-blocks %s' % l if l else ''} --blocks {k} --steps {max(3, 6144 // k)} --warmup 2 --ramp-steps {max(3, 10240 // k)}")
                  for (w, l) in (("c3", 0), ("i16", 0), ("i16r", 0), ("c3", 5.3), ("u4096", 0), ("c2", 0))
                  for k in (1024, 1536, 2048, 4096)
                  for (lab, env) in (("chained", ""), ("whole lists", "WBX_CHAIN=0"), ("grouped", "WBX_EXACT_MIN_BLOCKS=0"))
                  if not (k == 4096 and lab == "whole lists")]
    P["exact_k"] = [case(f"{w}{' L=%s' % l if l else ''} K={k} whole lists", "",
                         f"--workload {w} {'--clip-blocks %s' % l if l else ''} --blocks {k} --steps {6144 // k} --warmup 2 --ramp-steps {10240 // k}")
                    for (w, l) in (("i16r", 0), ("i16", 0), ("c3", 5.3), ("i24r", 0)) for k in (1024, 1536, 2048)]
    P["variants"] = [case(f"c3 variant={v}", f"WBX_MIX_VARIANT={v}", STD) for v in (24, 43, 82)]
    P["c2_groups"] = [case(f"c2 group={g} variant={v}", f"WBX_MIX_VARIANT={v}",
                           f"--workload c2 --group-size {g} --steps 20 --warmup 3 --ramp-steps 60")
                      for g in (128, 64, 32) for v in (43, 24)]
    P["alt"] = [case(f"{w} alt={a}", f"WBX_MIX_ALT={a}", f"--workload {w} {STD}") for w in ("c3", "c4") for a in (1, 0)]
    P["arena"] = [case(f"{w} {'per-clip allocations' if a else 'slabs'}", "WBX_CLIP_ARENA=0" if a else "",
                       f"--workload {w} {STD}") for w in ("c3", "c4") for a in (0, 1)]
    P["blocks"] = [case(f"F={f} {w} L={l} {v or 'default'}", v,
                        f"--workload {w} --block-frames {f} {'--clip-blocks %s' % l if l else ''} {STD}")
                   for f in (256, 1024) for w in ("c3", "i16") for l in (0, 5.3) for v in ("", "WBX_NO_CL2=1")]
    P["blocks_per_step"] = [case(f"K={k}", "", f"--blocks {k} --steps {5120 // k} --warmup 3 --ramp-steps {10240 // k}")
                            for k in (256, 512, 1024)]
    P["cl2"] = [case(f"{w} {'one channel/wave' if v else 'default'}", "WBX_NO_CL2=1" if v else "", f"--workload {w} {STD}")
                for w in ("c3", "c4", "i16", "mixfmt") for v in (0, 1)]
    P["lanes"] = [case(f"lanes={n} L={l}", f"WBX_PLAN_LANES={n}", f"--clip-blocks {l} --steps 10 --warmup 2 --ramp-steps 30")
                  for n in (64, 32, 16, 8) for l in (5.3, 20)]
    P["latency"] = [case(f"masked={m} no_uniform={nu}", f"WBX_MASKED_ROWS={m}" + (" WBX_NO_UNIFORM=1" if nu else ""),
                         "--steps 2 --warmup 1 --ramp-steps 2 --latency-blocks 400", lat=True)
                    for m in (1, 0) for nu in (0, 1)]
    P["latency_groups"] = [case(f"group={g}", "", f"--group-size {g} --steps 2 --warmup 1 --ramp-steps 2 --latency-blocks 400",
                                lat=True) for g in (128, 64, 32, 16, 8)]
    P["lean16"] = [case(f"i16r L={l} {v or 'default'}", v, f"--workload i16r {'--clip-blocks %s' % l if l else ''} {STD}")
                   for l in (0, 5.3) for v in ("", "WBX_NO_CL2=1", "WBX_NO_LEAN16=1")]
    P["masked"] = [case(f"{w} L={l} {'pre-render' if m else 'hot loop'}", "WBX_MASKED_ROWS=0" if m else "",
                        f"--workload {w} {'--clip-blocks %s' % l if l else ''} {STD}")
                   for w in ("c3", "i16") for l in (5.3, 20, 0) for m in (0, 1)]
    P["planprio"] = [case(f"L={l} plan_prio={p}", f"WBX_PLAN_PRIO={p}", f"--clip-blocks {l} --steps 10 --warmup 2 --ramp-steps 30")
                     for l in (5.3, 20) for p in ("hi", "lo")]
    P["ramp"] = [case(f"ramp={r}", "", f"--steps 20 --warmup 3 --ramp-steps {r}") for r in (40, 150, 400)]
    P["timer"] = [case(f"{w} {v or 'default (dispatch packet)'}", v, f"--workload {w} {STD}")
                  for w in ("c2", "c3") for v in ("", "WBX_TIMER_PACKETS=1", "WBX_KERNEL_TIMER=0")]
    P["uniform"] = [case(f"{w} no_uniform={nu}", "WBX_NO_UNIFORM=1" if nu else "", f"--workload {w} {STD}")
                    for w in ("c3", "i16r") for nu in (0, 1)]
    P["formats"] = [case(w, "", f"--workload {w} {STD}") for w in ("i16r", "i24r", "mixr", "mixfmt", "i16", "d96")]
    P["cuts"] = [case(f"{w} L={l}", "", f"--workload {w} {'--clip-blocks %s' % l if l else ''} {STD}")
                 for w in ("c3", "i16", "i16r", "i24r", "mixr") for l in (0, 5.3, 20)]
    # round 4: short blocks of sessions cut into clips — the packed masked-row instances (WBX_PACKED_X=1: on every shape,
    # measured with a full-length-chunk variant "2" as well, since removed) against the one-block-per-workgroup ones (0); "uncut*" = the uncut session forced
    # through the same instances (WBX_FORCE_CUT=1), "uncut" alone = the plain packed instance it normally takes
    P["packed"] = [case(f"F={f} uncut plain-packed", "", f"--block-frames {f} {STD}") for f in (128, 256)] + \
                  [case(f"F={f} {('L=%s' % l) if l else 'uncut*'} X={x}", f"WBX_PACKED_X={x}" + ("" if l else " WBX_FORCE_CUT=1"),
                        f"--block-frames {f} {'--clip-blocks %s' % l if l else ''} {STD}")
                   for f in (128, 256) for l in (0, 5.3, 40) for x in (0, 1)]
    P["packed1024"] = [case(f"F={f} L=5.3 K=1024 X={x}", f"WBX_PACKED_X={x}", f"--block-frames {f} --clip-blocks 5.3 --blocks 1024 {STD}")
                       for f in (128, 256) for x in (0, 1)]
    # round 4: the sequencer cut along the time axis (WBX_PLAN_SEG=0: one lane per track) on sessions cut into clips
    P["planseg"] = [case(f"{lab} seg={'on' if sg else 'off'}", "" if sg else "WBX_PLAN_SEG=0", args)
                    for (lab, args) in (("c3 L=5.3", f"--clip-blocks 5.3 {STD}"), ("c3 L=5.3 F=128", f"--block-frames 128 --clip-blocks 5.3 {STD}"),
                                        ("c3 L=5.3 F=256", f"--block-frames 256 --clip-blocks 5.3 {STD}"),
                                        ("c2 L=5.3", "--workload c2 --clip-blocks 5.3 --steps 20 --warmup 3 --ramp-steps 60"),
                                        ("c2 uncut", "--workload c2 --steps 20 --warmup 3 --ramp-steps 60"),
                                        ("i16r L=5.3", f"--workload i16r --clip-blocks 5.3 {STD}"), ("c3 L=20", f"--clip-blocks 20 {STD}"))
                    for sg in (1, 0)]
    if other_lib:   # head-to-head of two builds of libwbx.so
        P["lib"] = [case(f"{w} {'other' if o else 'this'}", f"WBX_LIB={other_lib}" if o else "", f"--workload {w} {STD}")
                    for w in ("c3", "c4", "i16r", "c2") for o in (0, 1)]
        P["lib_rev"] = [case(f"{w} {'other' if o else 'this'}", f"WBX_LIB={other_lib}" if o else "", f"--workload {w} {STD}")
                        for w in ("c4", "c2", "u4096") for o in (1, 0)]
    return P


def run_case(c, out):
    env = dict(os.environ)
    for kv in c["env"].split():
        k, v = kv.split("=", 1)
        env[k] = v
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + c["args"].split() + ["--no-cpu-baseline", "--no-configs"]
    if "--verify" in cmd:
        cmd.remove("--verify")      # (a case may ask for the oracle check of the rendered head; sweeps skip it)
    else:
        cmd.append("--no-verify")
    if not c["lat"]:
        cmd += ["--latency-blocks", "0"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        print(f"{c['label']:44s} FAILED rc={p.returncode} {p.stderr[-400:]!r}", flush=True)
        return
    d = json.loads(lines[-1])
    r = d["roofline"]
    msg = (f"{c['label']:44s} {d['value']:.4g} frames/s  step {d['ms_per_step']:.4f} ms  mix {r['kernel_ms_avg']:.4f} ms  "
           f"frac {r['frac']:.3f}  frac_step {r.get('frac_step', 0.0):.3f}  {r['kernel']}")
    if c["lat"] and d.get("latency_mode"):
        msg += f"  latency {d['latency_mode']['ms_per_block']:.4f} ms/block"
    if d.get("verify"):
        msg += f"  verify rms {d['verify'].get('rms'):.2e}"
    print(msg, flush=True)
    if out:
        d["ab_label"] = c["label"]
        d["ab_env"] = c["env"]
        out.write(json.dumps(d) + "\n")
        out.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("preset")
    ap.add_argument("cases", nargs="*")
    ap.add_argument("-r", "--repeats", type=int, default=2)
    ap.add_argument("-o", "--out", default=None)
    ap.add_argument("--other-lib", default=None)
    a = ap.parse_args()
    if a.preset == "run":
        cs = []
        for spec in a.cases:
            label, env, args = (spec.split(";") + ["", ""])[:3]
            cs.append(case(label.strip(), env.strip(), args.strip(), lat="--latency-blocks" in args))
    else:
        P = presets(a.other_lib)
        if a.preset not in P:
            raise SystemExit(f"unknown preset {a.preset}; have: {' '.join(sorted(P))} run")
        cs = P[a.preset]
    out = open(a.out, "a") if a.out else None
    for _ in range(a.repeats):
        for c in cs:
            run_case(c, out)


if __name__ == "__main__":
    main()
