set -u
O=gpurun_out/r03r; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
B="--no-cpu-baseline --no-configs --no-verify --latency-blocks 0"
one() { python bench.py $B "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-40s'%sys.argv[1], '%.4g'%d['value'], 'step %.4f mix %.4f frac %.3f frac_step %.3f'%(d['ms_per_step'], r['kernel_ms_avg'], r['frac'], r['frac_step']))" "$WBX_PLAN_BESIDE $*"; }
for PB in 1 0; do export WBX_PLAN_BESIDE=1 WBX_HOST_MASTER_DIRECT=$((1-PB))
 one --workload c3; one --workload c2; one --workload c4; one --workload i16r; one --workload i16
 one --clip-blocks 5.3; one --workload i16r --clip-blocks 5.3; one --blocks 256; one --blocks 256 --clip-blocks 5.3; one --workload c2 --blocks 256
done
export WBX_PLAN_BESIDE=1
R=$(pwd); cd /tmp
for n in c3 c2; do
 rocprofv3 --kernel-trace --output-format csv -d $R/$O/kt_$n -o $n -- python $R/bench.py $B --workload $n > /dev/null 2>&1
 echo "== $n"; python $R/tools/timeline.py $(find $R/$O/kt_$n -name "*kernel_trace.csv") 14
done
rocprofv3 --kernel-trace --output-format csv -d $R/$O/kt_c3_L -o c3 -- python $R/bench.py $B --clip-blocks 5.3 > /dev/null 2>&1
echo "== c3 L5.3"; python $R/tools/timeline.py $(find $R/$O/kt_c3_L -name "*kernel_trace.csv") 14
cd $R; python bench.py --no-cpu-baseline --no-configs > $O/bench.json 2>$O/bench.err; python -c "
import json
d=json.loads(open('$O/bench.json').readline()); print(d['value'], d['roofline']['frac'], d['roofline']['frac_step'], d['verify']['ok'], d.get('latency_mode'))"
