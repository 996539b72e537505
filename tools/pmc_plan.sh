#!/bin/bash
# PMC counters of plan_kernel on a cut session, run alone (WBX_OVERLAP=0); prints per-launch averages
L=${1:-5.3}
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; rm -rf /tmp/pmc_plan; mkdir -p /tmp/pmc_plan
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_SMEM SQ_WAIT_ANY"; do
  n=$(echo $grp | cut -d' ' -f1)
  WBX_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "plan_kernel" --pmc $grp --output-format csv -d /tmp/pmc_plan/$n -o p -- \
    python $R/bench.py --clip-blocks $L --steps 3 --warmup 1 --ramp-steps 2 --no-cpu-baseline --no-configs --latency-blocks 0 > /tmp/pmc_plan/$n.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_plan/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    v = v[1:] if len(v) > 1 else v
    print(f"{k:28s} launches {len(v):3d}  avg {sum(v)/len(v):14.1f}")
PY
