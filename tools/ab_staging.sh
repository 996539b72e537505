#!/bin/bash
# tools/ab_staging.sh — lane-per-track record staging (default build) against the cooperative copy of rounds 2-3
# (tools/_ab/libwbx_coop.so: the library built with -DWBX_LANE_STAGING=0), same box, alternating
cd "$(dirname "$0")/.."
run() {  # label, env..., -- bench args
  label=$1; shift
  python bench.py "$@" --steps 10 --no-configs --no-cpu-baseline --no-verify --latency-blocks 0 2>/dev/null | \
    python -c "import json,sys; l=json.loads(sys.stdin.read()); print('$label', l['roofline']['kernel'][16:], round(l['roofline']['kernel_ms_avg'],4), round(l['roofline']['frac'],3), round(l['roofline']['frac_step'],3))"
}
for rep in 1 2; do
for w in "--workload c3" "--workload c4" "--workload c3 --clip-blocks 5.3" "--workload i16r" "--workload i16 --clip-blocks 5.3" "--workload c2 --tracks 256" "--block-frames 128 --blocks 1024 --clip-blocks 5.3"; do
  run "lane  [$w]" $w
  WBX_LIB=$PWD/tools/_ab/libwbx_coop.so run "coop  [$w]" $w
done
done
