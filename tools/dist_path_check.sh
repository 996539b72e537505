R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3 4 5 6; do
for mode in "" "--force-dist-path"; do
python $R/bench.py $mode --no-cpu-baseline --latency-blocks 0 2>/dev/null > /tmp/o.json
python - "$mode" <<'PY'
import json,sys
d=json.loads([x for x in open('/tmp/o.json') if x.startswith('{')][-1])
print('%-18s step %.3f mix %.3f'%(sys.argv[1] or 'default', d['ms_per_step'], d['roofline']['kernel_ms_avg']))
PY
done; done
