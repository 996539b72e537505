#!/bin/bash
# A/B on one box: clip boundaries rendered in the hot loop (default) vs through the pre-render pass (WBX_MASKED_ROWS=0)
for L in 0 5.3 20; do
  for M in 1 0; do
    WBX_MASKED_ROWS=$M python bench.py --clip-blocks $L --steps 10 --warmup 2 --ramp-steps 30 --no-cpu-baseline --no-configs --latency-blocks 0 2>/dev/null | grep "^{" | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('L=$L masked=$M', '%.4g frames/s' % d['value'], 'step %.3f ms' % d['ms_per_step'], 'mix %.3f ms' % d['roofline']['kernel_ms_avg'], 'frac %.3f' % d['roofline']['frac'])"
  done
done
