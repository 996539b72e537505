#!/bin/bash
# interleaved A/B of library builds / env settings inside ONE gpurun call (same box, same thermal state)
# usage: tools/ab.sh "<label>=<ENV assignments>" ...   (workloads c3 c4, 2 rounds)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for spec in "$@"; do
    label=${spec%%=*}; envs=${spec#*=}
    for w in c3 c4; do
      env $envs python $R/bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --latency-blocks 0 > /tmp/ab.log 2>&1
      python - "$label" $w <<'PY'
import json,sys
try:
    l=[x for x in open("/tmp/ab.log") if x.startswith("{")][-1]; d=json.loads(l)
    print("%-14s %s  %.3e fr/s  step %.3f ms  mix %.3f ms  %.0f GB/s" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["achieved"]))
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e); print(open("/tmp/ab.log").read()[-600:])
PY
    done
  done
done
