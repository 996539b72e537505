#!/bin/bash
# repeat the default bench many times and print step / mix / tail per run: what does a slow run look like?
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-30}
for i in $(seq 1 $N); do
  python $R/bench.py --steps 10 --no-cpu-baseline --latency-blocks 0 2>/dev/null | python -c "
import sys,json
d=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); r=d['roofline']
print('run $i step %.3f mix %.3f tail %.3f other %.3f'%(d['ms_per_step'],r['kernel_ms_avg'],r['sum_tail_ms_avg'],d['ms_per_step']-r['kernel_ms_avg']-r['sum_tail_ms_avg']))"
done
