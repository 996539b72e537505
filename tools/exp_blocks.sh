#!/bin/bash
# c3 (and friends) at the block sizes a device back end really opens (period realigned to 32 frames, config.cpp:217-222):
# 480 = 10 ms at 48 kHz (WASAPI shared mode), 416 / 448 = 441 realigned, 960 = 20 ms — through the instances of the next shape
# above them (clone lanes, wbx_mix.h) and, WBX_RAGGED=0, through the general instance of earlier rounds.
# usage (GPU box): tools/exp_blocks.sh [tag] -> gpurun_out/<tag>/
set -u
O=gpurun_out/${1:-r05_blocks}; mkdir -p $O
B="--no-cpu-baseline --no-configs --latency-blocks 0"
for F in 480 416 960 448 320 1440; do
  timeout 300 python bench.py --block-frames $F $B > $O/bench_c3_F$F.json 2>> $O/err.log
  timeout 300 python bench.py --block-frames $F --clip-blocks 5.3 $B > $O/bench_c3_F${F}_L5.3.json 2>> $O/err.log
done
for F in 480 960; do
  WBX_RAGGED=0 timeout 300 python bench.py --block-frames $F $B > $O/bench_c3_F${F}_general.json 2>> $O/err.log
  WBX_RAGGED=0 timeout 300 python bench.py --block-frames $F --clip-blocks 5.3 $B > $O/bench_c3_F${F}_L5.3_general.json 2>> $O/err.log
done
for W in i16r i16 c4 i24r; do
  timeout 300 python bench.py --workload $W --block-frames 480 $B > $O/bench_${W}_F480.json 2>> $O/err.log
done
timeout 300 python bench.py $B > $O/bench_c3_F512.json 2>> $O/err.log
for f in $O/bench_*.json; do python tools/bench_line.py $(basename $f .json) < $f; done
