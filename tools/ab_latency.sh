#!/bin/bash
# the one-block callback path (Engine::process, K = 1): ms per 4096-track block
for M in 1 0; do for NU in "" 1; do
env WBX_MASKED_ROWS=$M ${NU:+WBX_NO_UNIFORM=1} python bench.py --steps 2 --warmup 1 --ramp-steps 2 --no-cpu-baseline --no-configs --latency-blocks 400 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('masked=$M no_uniform=${NU:-0}', 'latency %.4f ms/block' % d['latency_mode']['ms_per_block'])"
done; done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/lat; mkdir -p /tmp/lat
rocprofv3 --kernel-trace -d /tmp/lat -o lat -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --ramp-steps 2 --no-cpu-baseline --no-configs --latency-blocks 400 > /dev/null 2>&1
python - <<PY
import sqlite3, glob
db = glob.glob("/tmp/lat/**/*.db", recursive=True)[0]
con = sqlite3.connect(db); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = cur.execute(f"select {name}, start, end from kernels order by start").fetchall()
rows = rows[-1600:]       # the K = 1 blocks at the end
import collections
d = collections.defaultdict(list)
for n, s, e in rows: d[n.split("(")[0][:50]].append((e - s) / 1e3)
for k, v in d.items(): print(f"{k:52s} n={len(v):4d} avg {sum(v)/len(v):7.2f} us")
gaps = [(rows[i+1][1] - rows[i][2]) / 1e3 for i in range(len(rows) - 1)]
print("mean gap between consecutive kernels %.2f us" % (sum(gaps) / len(gaps)))
PY
