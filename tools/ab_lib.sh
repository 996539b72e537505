#!/bin/bash
# head-to-head of two builds of libwbx.so on one box over several workloads: tools/ab_lib.sh <other lib> [workloads...]
OTHER=$1; shift
WL=${@:-c3 i16r}
for W in $WL; do for L in "" "$OTHER" "" "$OTHER"; do
env ${L:+WBX_LIB=$L} python bench.py --workload $W --steps 20 --warmup 3 --ramp-steps 40 --no-cpu-baseline --no-configs --latency-blocks 0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W', '${L:+other}' or 'this ', '%.4g frames/s' % d['value'], 'step %.4f ms' % d['ms_per_step'], 'mix %.4f ms' % d['roofline']['kernel_ms_avg'], 'frac %.3f' % d['roofline']['frac'])"
done; done
