#!/bin/bash
# Profile bundle of a round (one gpurun call): gpu tests, the default bench line (headline + configs), the other
# workloads, the multi-GPU code path on one rank, rocprofv3 --kernel-trace --stats and separate --pmc passes for the
# configurations the bench line's `configs` object names (c3, c4, c3 cut into clips, i16r) and for the grouped order at
# 256-block renders.  Counters are never combined with trace domains other than --kernel-trace.
# usage (on the GPU box): tools/profile_round.sh [tag]      -> gpurun_out/<tag>/ ; then tools/refresh_profiles.sh <tag> <round>
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -n 4 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err
B="--no-cpu-baseline --no-configs"
for W in c4 c2 u4096 i16 i16r i24r d96 mixfmt mixr; do
  timeout 300 python bench.py --workload $W $B > $O/bench_$W.json 2>> $O/bench_default.err
done
for W in c3 i16 i16r; do for L in 5.3 20; do
  timeout 300 python bench.py --workload $W --clip-blocks $L $B > $O/bench_${W}_L$L.json 2>> $O/bench_default.err
done; done
# the grouped order (renders of 256 blocks: 128-track groups) for the same workloads
for W in c3 c4 c2 i16 i16r i24r mixr; do
  timeout 300 python bench.py --workload $W --blocks 256 $B > $O/bench_${W}_K256.json 2>> $O/bench_default.err
done
timeout 300 python bench.py --clip-blocks 5.3 --blocks 256 $B > $O/bench_c3_L5.3_K256.json 2>> $O/bench_default.err
# short blocks (the buffer sizes of a low-latency device) cut into clips: boundaries in the hot loop, and through the pre-render pass
for BF in 128 256; do
  timeout 300 python bench.py --block-frames $BF --blocks 1024 --clip-blocks 5.3 $B > $O/bench_c3_F${BF}_L5.3.json 2>> $O/bench_default.err
  WBX_MASKED_ROWS=0 timeout 300 python bench.py --block-frames $BF --blocks 1024 --clip-blocks 5.3 $B > $O/bench_c3_F${BF}_L5.3_prerender.json 2>> $O/bench_default.err
  timeout 300 python bench.py --block-frames $BF --blocks 1024 $B > $O/bench_c3_F${BF}.json 2>> $O/bench_default.err
done
# ... at the default render length (2048 blocks), and what the segmented sequencer is worth there (WBX_PLAN_SEG=0: one lane per track)
for BF in 128 256; do
  timeout 300 python bench.py --block-frames $BF --clip-blocks 5.3 $B > $O/bench_c3_F${BF}_L5.3_K2048.json 2>> $O/bench_default.err
  WBX_PLAN_SEG=0 timeout 300 python bench.py --block-frames $BF --clip-blocks 5.3 $B > $O/bench_c3_F${BF}_L5.3_K2048_noseg.json 2>> $O/bench_default.err
done
timeout 300 python bench.py --workload c2 --clip-blocks 5.3 $B > $O/bench_c2_L5.3.json 2>> $O/bench_default.err
WBX_PLAN_SEG=0 timeout 300 python bench.py --workload c2 --clip-blocks 5.3 $B > $O/bench_c2_L5.3_noseg.json 2>> $O/bench_default.err
WBX_PLAN_SEG=0 timeout 300 python bench.py --clip-blocks 5.3 $B > $O/bench_c3_L5.3_noseg.json 2>> $O/bench_default.err
( bash tools/seg_prof.sh ) > $O/seg_kernel_stats.txt 2>&1
for W in i24r mixr; do
  timeout 300 python bench.py --workload $W --clip-blocks 5.3 $B > $O/bench_${W}_L5.3.json 2>> $O/bench_default.err
  WBX_NO_FAM3=1 timeout 300 python bench.py --workload $W $B > $O/bench_${W}_fam1.json 2>> $O/bench_default.err
done
for M in reduce ordered chain; do
  timeout 300 python bench.py --force-dist-path --dist-mode $M $B > $O/bench_dist1_$M.json 2>> $O/bench_default.err
done
timeout 300 python tools/longrun_probe.py c3 2048 4 > $O/longrun_probe.txt 2>&1
# round 6: a 256-track session over 200 steps (a step is 0.4 ms: over 20 the one drain at the end is 4 % of the measurement), and
# what releasing the internal events to the device instead of the system is worth there (alternating)
for rep in 1 2 3; do
  timeout 200 python bench.py --workload c2 --steps 200 $B > $O/bench_c2_steps200_$rep.json 2>> $O/bench_default.err
  WBX_EVENT_SCOPE=system timeout 200 python bench.py --workload c2 --steps 200 $B > $O/bench_c2_steps200_sysscope_$rep.json 2>> $O/bench_default.err
done
cd /tmp
kt() {   # name, bench args...
  n=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -o $n -- python $R/bench.py $B --no-verify --latency-blocks 0 "$@" > $O/kt_${n}_bench.json 2> $O/kt_$n.err
}
kt c3
kt c4 --workload c4
kt c2 --workload c2
kt c3_L5.3 --clip-blocks 5.3
kt i16r --workload i16r
kt c3_K256 --blocks 256
pmc() {  # name, bench args...
  n=$1; shift
  mkdir -p $O/pmc_$n
  for grp in "FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "WRITE_SIZE TCC_HIT TCC_MISS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    g=$(echo $grp | cut -d' ' -f1)
    timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "mix_kernel" --pmc $grp --output-format csv -d $O/pmc_$n/$g -o $n -- \
      python $R/bench.py --steps 3 --warmup 1 --ramp-steps 3 $B --no-verify --latency-blocks 0 "$@" > $O/pmc_$n/$g.log 2>&1
  done
}
pmc c3
pmc c4 --workload c4
pmc c3_L5.3 --clip-blocks 5.3
pmc i16r --workload i16r
pmc c3_K256 --blocks 256
cd $R
# round 4: the one-block callback as one launch — kernel trace of 400 wbx_engine_process calls per session size (one dispatch
# each), the phases of the launch (tools/cb_clocks.py), and the same with the three launches of earlier rounds
cd /tmp
for N in 4096 64 8; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_callback_N$N -o cb -- python $R/tools/callback_calls.py $N 400 > $O/kt_callback_N$N.log 2>&1
done
cd $R
( for N in 4096 1024 256 64 8; do timeout 120 python tools/cb_clocks.py c3 $N; done ) > $O/callback_clocks.txt 2>&1
( echo "== one launch (default)"; timeout 300 python bench.py $B --no-verify --steps 3 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['latency_mode'])";
  echo "== three launches (WBX_CALLBACK_FUSED=0)"; WBX_CALLBACK_FUSED=0 timeout 300 python bench.py $B --no-verify --steps 3 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['latency_mode'])" ) > $O/callback_latency_ab.txt 2>&1
# rank processes that share the one GPU (RCCL's socket transport): world 2 and 8, chain mode (the default for >= 1024-block renders)
for Wd in 2 8; do
  WBX_SHARE_DEVICE=1 timeout 900 python bench.py --gpus $Wd --blocks 1024 --steps 4 --warmup 1 --ramp-steps 2 --session-blocks 2048 $B --latency-blocks 0 \
    > $O/bench_world${Wd}_shared_chain.json 2> $O/bench_world${Wd}_shared_chain.err
done
find $O -name "*.csv" -size +8M -delete
find $O -name "*.db" -delete
ls -R $O | head -60
