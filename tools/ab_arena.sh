#!/bin/bash
# clip storage in 1-GiB slabs (default) against one allocation per clip (WBX_CLIP_ARENA=0): the spread of the mix
# kernel's launch time from process to process
WL=${WL:-c3 c4}
for W in $WL; do for A in "" 0 "" 0 "" 0 "" 0 "" 0; do
env ${A:+WBX_CLIP_ARENA=0} python bench.py --workload $W --steps 20 --warmup 3 --ramp-steps 40 --no-cpu-baseline --no-configs --latency-blocks 0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W', '${A:+per-clip allocations}' or 'slabs               ', '%.4g frames/s' % d['value'], 'step %.4f ms' % d['ms_per_step'], 'mix %.4f ms' % d['roofline']['kernel_ms_avg'], 'frac %.3f' % d['roofline']['frac'])"
done; done
