#!/bin/bash
for G in 128 64 32 16; do
python bench.py --workload c2 --group-size $G --steps 20 --warmup 3 --ramp-steps 60 --no-cpu-baseline --no-configs --latency-blocks 0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 group=$G', '%.4g frames/s' % d['value'], 'step %.4f ms' % d['ms_per_step'], 'mix %.4f ms' % d['roofline']['kernel_ms_avg'], 'frac %.3f' % d['roofline']['frac'], 'tail %.3f' % d['roofline']['sum_tail_ms_avg'])"
done
