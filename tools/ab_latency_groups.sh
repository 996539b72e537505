#!/bin/bash
# the one-block callback path against the track-group size (tracks summed in order by one workgroup)
for G in 128 64 32 16 8; do
python bench.py --group-size $G --steps 2 --warmup 1 --ramp-steps 2 --no-cpu-baseline --no-configs --latency-blocks 400 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('group=$G', 'latency %.4f ms/block' % d['latency_mode']['ms_per_block'], 'K=256 step %.3f ms' % d['ms_per_step'])"
done
