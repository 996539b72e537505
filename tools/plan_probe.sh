#!/bin/bash
# tools/plan_probe.sh — how long does the sequencer of a session cut into clips take, and what does the step cost?
# usage: tools/plan_probe.sh <label> [env assignments...]   (runs c3 cut into 5.3-block clips and a 256-track cut session)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/prof
label=$1; shift
for wl in "c3 4096" "c2 256"; do
  set -- $wl "$@"; w=$1; n=$2; shift 2
  out=gpurun_out/prof/plan_${label}_${w}
  rm -rf $out
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python bench.py --workload $w --tracks $n --clip-blocks 5.3 --steps 10 --warmup 2 \
      --ramp-steps 4 --no-configs --no-cpu-baseline --no-verify --latency-blocks 0 > $out.json 2> $out.err
  python - "$out" "$label" "$w" <<'PY'
import csv, glob, json, sys
out, label, w = sys.argv[1:4]
line = json.load(open(out + ".json"))
rows = []
for f in glob.glob(out + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
pick = lambda name: [r for r in rows if name in r["Name"]]
msg = f"{label:18s} {w}: step {line['ms_per_step']:.3f} ms  frac_step {line['roofline']['frac_step']:.3f}"
for name in ("plan_kernel", "mix_kernel", "gen_kernel"):
    for r in pick(name):
        msg += f" | {r['Name'].split('(')[0][-28:]} avg {float(r['AverageNs']) / 1e6:.3f} ms x{r['Calls']}"
print(msg)
PY
done
