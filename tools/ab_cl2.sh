#!/bin/bash
# both channels per lane (mix_kernel<..., CL = 2>, the default for resampling stereo 512-frame sessions) against the
# one-channel-per-wave instances (WBX_NO_CL2=1): tools/ab_cl2.sh ; WL="c3 i16" tools/ab_cl2.sh
WL=${WL:-c3 c4 i16 mixfmt}
for W in $WL; do for V in "" 1 "" 1; do
env ${V:+WBX_NO_CL2=$V} python bench.py --workload $W --steps 20 --warmup 3 --ramp-steps 40 --no-cpu-baseline --no-configs 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W', '${V:+one channel/wave }' or 'default          ', '%.4g frames/s' % d['value'], 'step %.4f ms' % d['ms_per_step'], 'mix %.4f ms' % d['roofline']['kernel_ms_avg'], 'frac %.3f' % d['roofline']['frac'], 'latency %.4f ms' % d['latency_mode']['ms_per_block'], d['roofline']['kernel'])"
done; done
for V in "" 1023 "" 1023; do
env ${V:+WBX_MIX_VARIANT=$V} python bench.py --workload i16 --steps 20 --warmup 3 --ramp-steps 40 --no-cpu-baseline --no-configs --latency-blocks 0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('i16 variant=${V:-default}', '%.4g frames/s' % d['value'], 'step %.4f ms' % d['ms_per_step'], 'mix %.4f ms' % d['roofline']['kernel_ms_avg'], 'frac %.3f' % d['roofline']['frac'])"
done
