#!/bin/bash
# c2 (256 tracks): track-group size x mix kernel instance (WBX_MIX_VARIANT = 10*U + W)
for G in 128 64 32; do for V in 43 24 44 16; do
WBX_MIX_VARIANT=$V python bench.py --workload c2 --group-size $G --steps 20 --warmup 3 --ramp-steps 60 --no-cpu-baseline --no-configs --latency-blocks 0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 group=$G variant=$V', '%.4g frames/s' % d['value'], 'step %.4f ms' % d['ms_per_step'], 'mix %.4f ms' % d['roofline']['kernel_ms_avg'], 'frac %.3f' % d['roofline']['frac'], 'tail %.3f' % d['roofline']['sum_tail_ms_avg'])"
done; done
