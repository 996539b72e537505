set -u
O=gpurun_out/r03p; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
B="--no-cpu-baseline --no-configs --no-verify --latency-blocks 0"
for W in c3 c2 c4 i16r; do
 python bench.py $B --workload $W 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$W', '%.4g'%d['value'], 'step %.4f mix %.4f frac %.3f frac_step %.3f'%(d['ms_per_step'], r['kernel_ms_avg'], r['frac'], r['frac_step']))"
done
python bench.py $B --blocks 256 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('c3 K256', '%.4g'%d['value'], 'step %.4f mix %.4f frac %.3f frac_step %.3f'%(d['ms_per_step'], r['kernel_ms_avg'], r['frac'], r['frac_step']))"
R=$(pwd); cd /tmp
for n in c3 c2; do
 rocprofv3 --kernel-trace --output-format csv -d $R/$O/kt_$n -o $n -- python $R/bench.py $B --workload $n > /dev/null 2>&1
 echo "== $n"; python $R/tools/timeline.py $(find $R/$O/kt_$n -name "*kernel_trace.csv") 14
done
rocprofv3 --kernel-trace --output-format csv -d $R/$O/kt_c3_256 -o c3 -- python $R/bench.py $B --blocks 256 > /dev/null 2>&1
echo "== c3 K256"; python $R/tools/timeline.py $(find $R/$O/kt_c3_256 -name "*kernel_trace.csv") 14
cd $R; python bench.py --no-cpu-baseline --no-configs > $O/bench.json 2>$O/bench.err; python -c "
import json
d=json.loads(open('$O/bench.json').readline()); print(d['value'], d['roofline']['frac'], d['roofline']['frac_step'], d['verify']['ok'], d.get('latency_mode'))"
