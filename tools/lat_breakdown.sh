#!/bin/bash
# per-kernel durations of the one-block callback path for a given track-group size
G=${1:-64}
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/lat; mkdir -p /tmp/lat
rocprofv3 --kernel-trace -d /tmp/lat -o lat -- python $GRAFT_REPO_ROOT/bench.py --group-size $G --steps 2 --warmup 1 --ramp-steps 2 --no-cpu-baseline --no-configs --latency-blocks 400 > /tmp/lat/bench.log 2>&1
grep "^{" /tmp/lat/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('group=$G latency %.4f ms/block' % d['latency_mode']['ms_per_block'])"
python - <<PY
import sqlite3, glob, collections
db = glob.glob("/tmp/lat/**/*.db", recursive=True)[0]
con = sqlite3.connect(db); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = cur.execute(f"select {name}, start, end from kernels order by start").fetchall()
rows = rows[-900:]
d = collections.defaultdict(list)
for n, s, e in rows: d[n.split("(")[0][:50]].append((e - s) / 1e3)
for k, v in d.items(): print(f"  {k:52s} n={len(v):4d} avg {sum(v)/len(v):7.2f} us")
per = collections.defaultdict(list)
# time from plan start to sum end per block, and from sum end to next plan start
plans = [(s, e) for n, s, e in rows if "plan_kernel" in n]
sums = [(s, e) for n, s, e in rows if "sum_kernel" in n]
if len(plans) > 10 and len(sums) > 10:
    n = min(len(plans), len(sums))
    sp = [(sums[i][1] - plans[i][0]) / 1e3 for i in range(n) if sums[i][1] > plans[i][0]]
    gp = [(plans[i + 1][0] - sums[i][1]) / 1e3 for i in range(n - 1) if plans[i + 1][0] > sums[i][1]]
    print("  plan start -> sum end %.2f us   sum end -> next plan start %.2f us" % (sum(sp) / len(sp), sum(gp) / len(gp)))
PY
