#!/bin/bash
# mix_kernel<U, true, W, false, 1> instances on c3: WBX_MIX_VARIANT = 10*U + W
for V in 24 43 82 24; do
WBX_MIX_VARIANT=$V python bench.py --steps 20 --warmup 3 --ramp-steps 40 --no-cpu-baseline --no-configs --latency-blocks 0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant=$V', '%.4g frames/s' % d['value'], 'step %.4f ms' % d['ms_per_step'], 'mix %.4f ms' % d['roofline']['kernel_ms_avg'], 'frac %.3f' % d['roofline']['frac'])"
done
