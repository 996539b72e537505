#!/bin/bash
# head-to-head of two builds of libwbx.so on one box: WBX_LIB=<other build> (tools/ab_head.sh /path/to/other/libwbx.so)
OTHER=$1
for W in c3 c4; do for L in "" "$OTHER" "" "$OTHER"; do
env ${L:+WBX_LIB=$L} python bench.py --workload $W --steps 20 --warmup 3 --ramp-steps 40 --no-cpu-baseline --no-configs --latency-blocks 0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W', '${L:+other}' or 'this ', '%.4g frames/s' % d['value'], 'step %.4f ms' % d['ms_per_step'], 'mix %.4f ms' % d['roofline']['kernel_ms_avg'], 'frac %.3f' % d['roofline']['frac'])"
done; done
