#!/bin/bash
# what the HIP-event kernel timer costs per step (WBX_KERNEL_TIMER=0 records no timing events)
for T in 1 0 1 0; do
WBX_KERNEL_TIMER=$T python bench.py --steps 20 --warmup 3 --ramp-steps 40 --no-cpu-baseline --no-configs --latency-blocks 0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('timer=$T', '%.4g frames/s' % d['value'], 'step %.4f ms' % d['ms_per_step'], 'mix %.4f ms' % d['roofline']['kernel_ms_avg'])"
done
