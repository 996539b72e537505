#!/bin/bash
# blocks per step (one device pass): the launch-to-launch gap of ~45 us is per step
for K in 256 512 1024 256 512 1024 256 512 1024; do
python bench.py --blocks $K --steps $((5120 / K)) --warmup 3 --ramp-steps $((10240 / K)) --no-cpu-baseline --no-configs --latency-blocks 0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=$K', '%.4g frames/s' % d['value'], 'step %.4f ms' % d['ms_per_step'], 'per 256 blocks %.4f ms' % (d['ms_per_step'] * 256 / $K), 'mix %.4f ms' % d['roofline']['kernel_ms_avg'], 'frac %.3f' % d['roofline']['frac'])"
done
