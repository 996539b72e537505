#!/bin/bash
# clip boundaries in the hot loop (default) against the pre-render pass (WBX_MASKED_ROWS=0), fp32 resampled (c3) and
# 16-bit PCM at the session rate (i16), sessions cut into clips of L blocks
for W in c3 i16; do for L in 5.3 20 0; do for M in "" 0; do
env ${M:+WBX_MASKED_ROWS=$M} python bench.py --workload $W $( [ $L != 0 ] && echo --clip-blocks $L ) --steps 20 --warmup 3 --ramp-steps 40 --no-cpu-baseline --no-configs --latency-blocks 0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W L=$L', '${M:+pre-render }' or 'hot loop   ', '%.4g frames/s' % d['value'], 'step %.4f ms' % d['ms_per_step'], 'mix %.4f ms' % d['roofline']['kernel_ms_avg'], 'frac %.3f' % d['roofline']['frac'], d['roofline']['kernel'])"
done; done; done
