#!/bin/bash
# the kernel timer's events on the mix kernel's own dispatch packet (default) against an event record either side of
# the launch (WBX_TIMER_PACKETS=1) and no timer at all (WBX_KERNEL_TIMER=0)
for W in c2 c3; do for V in "" "WBX_TIMER_PACKETS=1" "WBX_KERNEL_TIMER=0" "" "WBX_TIMER_PACKETS=1" "WBX_KERNEL_TIMER=0"; do
env $V python bench.py --workload $W --steps 20 --warmup 3 --ramp-steps 40 --no-cpu-baseline --no-configs --latency-blocks 0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W', '${V:-default (on the dispatch packet)}', '%.4g frames/s' % d['value'], 'step %.4f ms' % d['ms_per_step'], 'mix %.4f ms' % d['roofline']['kernel_ms_avg'], 'frac %.3f' % d['roofline']['frac'])"
done; done
