#!/bin/bash
# repeat the default bench many times and print step / mix / tail / slowest host enqueue per run
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-30}
for i in $(seq 1 $N); do
  python $R/bench.py --steps 20 --no-cpu-baseline --latency-blocks 0 2>/dev/null > /tmp/oh.json
  python - $i <<'PY'
import sys, json
d = json.loads([x for x in open('/tmp/oh.json') if x.startswith('{')][-1]); r = d['roofline']
print('run %s step %.3f mix %.3f tail %.3f enq_max %.3f' % (sys.argv[1], d['ms_per_step'], r['kernel_ms_avg'], r['sum_tail_ms_avg'], d['host_enqueue_ms_max']))
PY
done
