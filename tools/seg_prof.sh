R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/segprof; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for V in beside roomy; do
  E=""; [ $V = roomy ] && E="WBX_PLAN_BESIDE=0"
  env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$V -o run -- python $R/bench.py --block-frames 128 --clip-blocks 5.3 --steps 10 --warmup 2 --ramp-steps 10 --no-cpu-baseline --no-configs --no-verify --latency-blocks 0 > $O/$V.json 2> $O/$V.err
  echo "== $V"; python - "$O/$V" <<'PY'
import csv, glob, sys, json
out=sys.argv[1]
d=json.loads([l for l in open(out+".json") if l.startswith("{")][-1])
print("step", d["ms_per_step"], "mix", d["roofline"]["kernel_ms_avg"], "frac_step", d["roofline"]["frac_step"])
for f in glob.glob(out+"/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("plan","mix_kernel","sum_kernel","times_copy","gen_kernel")):
            print(f"  {r['Name'].split('(')[0][-50:]:52s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e6:.3f} ms  min {float(r['MinNs'])/1e6:.3f} max {float(r['MaxNs'])/1e6:.3f}")
PY
done
