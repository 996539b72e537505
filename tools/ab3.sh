R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { label=$1; shift; envs=$1; shift
  env $envs python $R/bench.py --workload c3 "$@" --no-cpu-baseline --latency-blocks 0 > /tmp/ab.log 2>&1
  python - "$label" <<'PY'
import json,sys
try:
    l=[x for x in open("/tmp/ab.log") if x.startswith("{")][-1]; d=json.loads(l)
    print("%-22s %.3e fr/s  step %.3f ms  mix %.3f ms  %.0f GB/s" % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["achieved"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open("/tmp/ab.log").read()[-600:])
PY
}
for rep in 1 2; do
 for v in 16 24 43; do
  run "s64 g64 v$v" "WBX_MIX_VARIANT=$v"
  run "s64 g128 v$v" "WBX_MIX_VARIANT=$v" --group-size 128
  run "s128 g128 v$v" "WBX_MIX_VARIANT=$v WBX_LIB=whitebox_amd/ab/libwbx_s128.so" --group-size 128
 done
done
