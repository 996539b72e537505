#!/bin/bash
# run the default bench under rocprofv3 kernel-trace until a slow run (ms_per_step > 1.05) shows up; keep its trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
mkdir -p $R/gpurun_out/outlier
for i in $(seq 1 ${1:-14}); do
  rm -rf /tmp/ot
  rocprofv3 --kernel-trace --output-format csv -d /tmp/ot -o c3 -- python $R/bench.py --no-cpu-baseline --latency-blocks 0 > /tmp/ot.json 2>/dev/null
  ms=$(python -c "
import json
d=json.loads([x for x in open('/tmp/ot.json') if x.startswith('{')][-1]); print('%.3f %.3f'%(d['ms_per_step'], d['roofline']['kernel_ms_avg']))")
  echo "run $i step/mix $ms"
  slow=$(python -c "print(1 if float('$ms'.split()[0])>1.05 else 0)")
  if [ "$slow" = "1" ]; then
    python - <<'PY' > $R/gpurun_out/outlier/timeline.txt
import csv,glob
f=glob.glob('/tmp/ot/**/*kernel_trace.csv', recursive=True)[0]
ev=[]
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    k='mix' if 'mix_kernel' in n else 'plan' if 'plan_kernel' in n else 'sum' if 'sum_kernel' in n else 'gen' if 'gen_kernel' in n else ('fill' if 'fillBuffer' in n else ('copy' if 'copyBuffer' in n else None))
    if k: ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),k))
ev.sort(); t0=ev[0][0]
for s,e,k in ev[-140:]:
    print('%-5s start %9.1f  end %9.1f  dur %7.1f us'%(k,(s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3))
PY
    echo "captured slow run"; break
  fi
done
