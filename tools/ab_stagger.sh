#!/bin/bash
# start offset of each clip inside its allocation (WBX_CLIP_STAGGER = step in bytes, offsets (i * step) mod 64 KiB): HBM
# channel interleave of the rows a workgroup has in flight
WL=${WL:-c3}
for W in $WL; do for S in 0 1280 4352 768 0 1280 4352 768 0 1280 4352 768; do
WBX_CLIP_STAGGER=$S python bench.py --workload $W --steps 20 --warmup 3 --ramp-steps 40 --no-cpu-baseline --no-configs --latency-blocks 0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W stagger=$S', '%.4g frames/s' % d['value'], 'step %.4f ms' % d['ms_per_step'], 'mix %.4f ms' % d['roofline']['kernel_ms_avg'], 'frac %.3f' % d['roofline']['frac'])"
done; done
