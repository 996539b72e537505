#!/bin/bash
for L in 5.3 20; do for P in hi lo hi lo; do
WBX_PLAN_PRIO=$P python bench.py --clip-blocks $L --steps 10 --warmup 2 --ramp-steps 30 --no-cpu-baseline --no-configs --latency-blocks 0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('L=$L plan_prio=$P', '%.4g frames/s' % d['value'], 'step %.3f ms' % d['ms_per_step'], 'mix %.3f ms' % d['roofline']['kernel_ms_avg'])"
done; done
