#!/bin/bash
# all-16-bit sessions: the lean 16-bit family (mix_kernel<.., FAM = 2, ..>, both channels per lane / one channel per wave)
# against the everything instance (WBX_NO_LEAN16=1)
for W in i16r; do for L in 0 5.3; do for V in "" "WBX_NO_CL2=1" "WBX_NO_LEAN16=1" "" "WBX_NO_CL2=1" "WBX_NO_LEAN16=1"; do
env $V python bench.py --workload $W $( [ $L != 0 ] && echo --clip-blocks $L ) --steps 20 --warmup 3 --ramp-steps 40 --no-cpu-baseline --no-configs --latency-blocks 0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W L=$L', '${V:-default}', '%.4g frames/s' % d['value'], 'step %.4f ms' % d['ms_per_step'], 'mix %.4f ms' % d['roofline']['kernel_ms_avg'], 'frac %.3f' % d['roofline']['frac'], d['roofline']['kernel'])"
done; done; done
