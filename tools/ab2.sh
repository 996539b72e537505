#!/bin/bash
# like ab.sh but with explicit bench args: tools/ab2.sh "<bench args>" "<label>=<ENV>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
BA=$1; shift
for rep in 1 2; do
  for spec in "$@"; do
    label=${spec%%=*}; envs=${spec#*=}
    for w in c3 c4; do
      env $envs python $R/bench.py --workload $w $BA --no-cpu-baseline --latency-blocks 0 > /tmp/ab.log 2>&1
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
