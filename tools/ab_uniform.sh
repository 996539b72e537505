#!/bin/bash
# hoisted position products for one-resampling-ratio sessions (default) vs per-row fp64 multiplies (WBX_NO_UNIFORM=1)
for W in c3 i16r; do for NU in "" 1 "" 1; do
env ${NU:+WBX_NO_UNIFORM=1} python bench.py --workload $W --steps 20 --warmup 3 --ramp-steps 40 --no-cpu-baseline --no-configs --latency-blocks 0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W no_uniform=${NU:-0}', '%.4g frames/s' % d['value'], 'step %.4f ms' % d['ms_per_step'], 'mix %.4f ms' % d['roofline']['kernel_ms_avg'], 'frac %.3f' % d['roofline']['frac'])"
done; done
