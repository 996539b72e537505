#!/bin/bash
# Round-2 profile bundle (one gpurun call): gpu tests, the default bench line (headline + configs), the other
# workloads, the multi-GPU code path on one rank, rocprofv3 --kernel-trace --stats and separate --pmc passes for the
# four configurations the bench line's `configs` object names (c3, c4, c3 cut into clips, i16r).
# usage (on the GPU box): tools/profile_r02.sh [tag]
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for W in c4 c2 i16 i16r i24r d96 mixfmt mixr; do
  timeout 300 python bench.py --workload $W --no-cpu-baseline --no-configs > $O/bench_$W.json 2>> $O/bench_default.err
done
timeout 300 python bench.py --clip-blocks 5.3 --no-cpu-baseline --no-configs > $O/bench_c3_L5.3.json 2>> $O/bench_default.err
timeout 300 python bench.py --clip-blocks 20 --no-cpu-baseline --no-configs > $O/bench_c3_L20.json 2>> $O/bench_default.err
timeout 300 python bench.py --workload i16 --clip-blocks 5.3 --no-cpu-baseline --no-configs > $O/bench_i16_L5.3.json 2>> $O/bench_default.err
timeout 300 python bench.py --workload i16 --clip-blocks 20 --no-cpu-baseline --no-configs > $O/bench_i16_L20.json 2>> $O/bench_default.err
timeout 300 python bench.py --workload i16r --clip-blocks 5.3 --no-cpu-baseline --no-configs > $O/bench_i16r_L5.3.json 2>> $O/bench_default.err
timeout 300 python bench.py --workload i16r --clip-blocks 20 --no-cpu-baseline --no-configs > $O/bench_i16r_L20.json 2>> $O/bench_default.err
timeout 300 python bench.py --force-dist-path --no-cpu-baseline --no-configs > $O/bench_dist1_reduce.json 2>> $O/bench_default.err
timeout 300 python bench.py --force-dist-path --dist-mode ordered --no-cpu-baseline --no-configs > $O/bench_dist1_ordered.json 2>> $O/bench_default.err
cd /tmp
kt() {   # name, bench args...
  n=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -o $n -- python $R/bench.py --no-cpu-baseline --no-configs --latency-blocks 0 "$@" > $O/kt_${n}_bench.json 2> $O/kt_$n.err
}
kt c3
kt c4 --workload c4
kt c3_L5.3 --clip-blocks 5.3
kt i16r --workload i16r
WBX_OVERLAP=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3_noov -o c3 -- python $R/bench.py --no-cpu-baseline --no-configs --latency-blocks 0 > $O/kt_c3_noov_bench.json 2> $O/kt_c3_noov.err
pmc() {  # name, bench args...
  n=$1; shift
  for grp in "FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "WRITE_SIZE TCC_HIT TCC_MISS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    g=$(echo $grp | cut -d' ' -f1)
    timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "mix_kernel" --pmc $grp --output-format csv -d $O/pmc_$n/$g -o $n -- \
      python $R/bench.py --steps 3 --warmup 1 --ramp-steps 4 --no-cpu-baseline --no-configs --latency-blocks 0 "$@" > $O/pmc_$n/$g.log 2>&1
  done
}
mkdir -p $O/pmc_c3 $O/pmc_c4 $O/pmc_c3_L5.3 $O/pmc_i16r
pmc c3
pmc c4 --workload c4
pmc c3_L5.3 --clip-blocks 5.3
pmc i16r --workload i16r
cd $R
find $O -name "*.csv" -size +8M -delete
find $O -name "*.db" -delete
ls -R $O | head -80
