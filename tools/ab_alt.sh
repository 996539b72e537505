#!/bin/bash
# consecutive mixes on alternating streams (default) vs all on the main stream (WBX_MIX_ALT=0)
for W in c3 c4; do for A in 1 0 1 0; do
WBX_MIX_ALT=$A python bench.py --workload $W --steps 20 --warmup 3 --ramp-steps 40 --no-cpu-baseline --no-configs --latency-blocks 0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W alt=$A', '%.4g frames/s' % d['value'], 'step %.4f ms' % d['ms_per_step'], 'mix %.4f ms' % d['roofline']['kernel_ms_avg'], 'frac %.3f' % d['roofline']['frac'])"
done; done
