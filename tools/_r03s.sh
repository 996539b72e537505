set -u
O=gpurun_out/r03t; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; grep -E 'passed|failed' $O/pytest.txt | tail -2
B="--no-cpu-baseline --no-configs --no-verify --latency-blocks 0"
one() { python bench.py $B "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-40s'%sys.argv[1], '%.4g'%d['value'], 'step %.4f mix %.4f frac %.3f frac_step %.3f enq_max %.3f ms'%(d["ms_per_step"], r["kernel_ms_avg"], r["frac"], r["frac_step"], d["host_enqueue_ms_max"]))" "$*"; }
one --workload c3; one --workload c2; one --workload c4; one --workload i16r; one --workload i16
one --clip-blocks 5.3; one --workload i16r --clip-blocks 5.3; one --blocks 256; one --blocks 1024; one --blocks 4096; one --blocks 4096 --clip-blocks 5.3; one --workload c2 --blocks 256; one --workload c2 --blocks 1024
R=$(pwd); cd /tmp
for n in c3 c2; do
 rocprofv3 --kernel-trace --output-format csv -d $R/$O/kt_$n -o $n -- python $R/bench.py $B --workload $n > /dev/null 2>&1
 echo "== $n"; python $R/tools/timeline.py $(find $R/$O/kt_$n -name "*kernel_trace.csv") 14
done
cd $R; python bench.py > $O/bench.json 2>$O/bench.err; python -c "
import json
d=json.loads(open('$O/bench.json').readline()); print(d['value'], d['roofline']['frac'], d['roofline']['frac_step'], d['verify']['ok'], d.get('latency_mode'))
for k,v in d['configs'].items(): print(k, '%.4g'%v['value'], v['roofline']['frac'], v['roofline']['frac_step'], (v.get('verify') or {}).get('ok'))"
