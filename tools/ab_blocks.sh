#!/bin/bash
# 256- and 1024-frame stereo blocks: both channels per lane (default when resampled / integer clips exist) against the
# one-channel-per-wave instances (WBX_NO_CL2=1), uncut and cut into clips
for F in 256 1024; do for W in c3 i16; do for L in 0 5.3; do for V in "" "WBX_NO_CL2=1"; do
env $V python bench.py --workload $W --block-frames $F $( [ $L != 0 ] && echo --clip-blocks $L ) --steps 20 --warmup 3 --ramp-steps 40 --no-cpu-baseline --no-configs --latency-blocks 0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('F=$F $W L=$L', '${V:-default      }', '%.4g frames/s' % d['value'], 'step %.4f ms' % d['ms_per_step'], 'mix %.4f ms' % d['roofline']['kernel_ms_avg'], 'frac %.3f' % d['roofline']['frac'], d['roofline']['kernel'])"
done; done; done; done
