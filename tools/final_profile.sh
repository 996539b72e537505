#!/bin/bash
# Round profile bundle: gpu tests, default bench line, rocprofv3 kernel-trace stats (overlap on and off), PMC passes.
# usage (on the GPU box): tools/final_profile.sh <tag>
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --workload c4 --no-cpu-baseline > $O/bench_c4.json 2>> $O/bench_default.err
timeout 300 python bench.py --workload i16 --no-cpu-baseline > $O/bench_i16.json 2>> $O/bench_default.err
timeout 300 python bench.py --workload c2 --no-cpu-baseline > $O/bench_c2.json 2>> $O/bench_default.err
timeout 300 python bench.py --workload d96 --no-cpu-baseline > $O/bench_d96.json 2>> $O/bench_default.err
timeout 300 python bench.py --workload i16r --no-cpu-baseline > $O/bench_i16r.json 2>> $O/bench_default.err
timeout 300 python bench.py --workload i24r --no-cpu-baseline > $O/bench_i24r.json 2>> $O/bench_default.err
timeout 300 python bench.py --workload mixfmt --no-cpu-baseline > $O/bench_mixfmt.json 2>> $O/bench_default.err
timeout 300 python bench.py --workload mixr --no-cpu-baseline > $O/bench_mixr.json 2>> $O/bench_default.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o c3 -- python $R/bench.py --no-cpu-baseline --latency-blocks 0 > $O/kt_bench.json 2> $O/kt.err
WBX_OVERLAP=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_noov -o c3 -- python $R/bench.py --no-cpu-baseline --latency-blocks 0 > $O/kt_noov_bench.json 2> $O/kt_noov.err
cd $R
bash tools/pmc_run.sh c3 $O/pmc
find $O -name "*.csv" -size +8M -delete
ls -R $O | head -60
