#!/bin/bash
# run-to-run spread of the headline against the number of untimed ramp steps in front of the timed ones
for R in 40 150 400 40 150 400 40 150 400; do
python bench.py --steps 20 --warmup 3 --ramp-steps $R --no-cpu-baseline --no-configs --latency-blocks 0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ramp=$R', '%.4g frames/s' % d['value'], 'step %.4f ms' % d['ms_per_step'], 'mix %.4f ms' % d['roofline']['kernel_ms_avg'], 'frac %.3f' % d['roofline']['frac'])"
done
