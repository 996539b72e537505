#!/bin/bash
# rocprofv3 kernel trace of the cut-into-clips session; prints the per-kernel table (tools/rocpd_stats.py)
L=${1:-5.3}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_cut; mkdir -p /tmp/prof_cut
rocprofv3 --kernel-trace -d /tmp/prof_cut -o cut -- python $GRAFT_REPO_ROOT/bench.py --clip-blocks $L --steps 10 --warmup 2 --ramp-steps 30 --no-cpu-baseline --no-configs --latency-blocks 0 "$@" > /tmp/prof_cut/bench.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/prof_cut -name "*.db" | head -1)
