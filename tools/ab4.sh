R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { label=$1; shift; envs=$1; shift
  env $envs python $R/bench.py "$@" --no-cpu-baseline --latency-blocks 0 > /tmp/ab.log 2>&1
  python - "$label" <<'PY'
import json,sys
try:
    l=[x for x in open("/tmp/ab.log") if x.startswith("{")][-1]; d=json.loads(l)
    print("%-22s %.3e fr/s  step %.3f ms  mix %.3f ms  %.0f GB/s" % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["achieved"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open("/tmp/ab.log").read()[-600:])
PY
}
WBX_LIB=whitebox_amd/ab/libwbx_s256.so python -m pytest $R/tests -m gpu -x -q 2>&1 | tail -1
for rep in 1 2 3; do
  run "c3 s64 g64" "A=1" --workload c3
  run "c3 s128 g128" "WBX_LIB=whitebox_amd/ab/libwbx_s128.so" --workload c3 --group-size 128
  run "c3 s256 g256" "WBX_LIB=whitebox_amd/ab/libwbx_s256.so" --workload c3 --group-size 256
  run "c3 s256 g128" "WBX_LIB=whitebox_amd/ab/libwbx_s256.so" --workload c3 --group-size 128
  run "c4 s256" "WBX_LIB=whitebox_amd/ab/libwbx_s256.so" --workload c4
done
