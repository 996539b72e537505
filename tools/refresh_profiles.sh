#!/bin/bash
# Turn the output of tools/profile_round.sh <tag> (gpurun_out/<tag>/, scratch) into the committed files under profiles/.
# usage (in the repo root): tools/refresh_profiles.sh <tag> [round prefix, default r03]
set -e
TAG=${1:?tag}
RN=${2:-r04}
O=gpurun_out/$TAG
for f in $O/bench_*.json; do W=$(basename $f .json); W=${W#bench_}; grep "^{" $f | tail -1 > profiles/${RN}_bench_$W.json; done
cp $O/pytest_gpu.log profiles/${RN}_pytest_gpu.log
cp $O/longrun_probe.txt profiles/${RN}_longrun_probe.txt
[ -f $O/seg_kernel_stats.txt ] && cp $O/seg_kernel_stats.txt profiles/${RN}_seg_kernel_stats.txt
kof() { python - "$1" <<'PY'
import json, sys
print(json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])["config"]["blocks_per_step"])
PY
}
NAMES="c3 c4 c2 c3_L5.3 i16r c3_K256"
for n in $NAMES; do
  K=$(kof $O/kt_${n}_bench.json)
  b=${n%_K256}
  cp $O/kt_$n/${n}_kernel_stats.csv profiles/${RN}_${b}_K${K}_kernel_stats_rocprofv3.csv
  grep "^{" $O/kt_${n}_bench.json | tail -1 > profiles/${RN}_${b}_K${K}_bench_line_of_profiled_run.json
  if [ -d $O/pmc_$n ]; then python tools/pmc_summary.py $O/pmc_$n mix_kernel > profiles/${RN}_${b}_K${K}_mix_pmc.txt; fi
done
C="python bench.py --no-cpu-baseline --no-configs --no-verify --latency-blocks 0"
python tools/kernel_summary.py \
  $O/kt_c3/c3_kernel_trace.csv $O/kt_c3_bench.json "$C" \
  $O/kt_c4/c4_kernel_trace.csv $O/kt_c4_bench.json "$C --workload c4" \
  $O/kt_c2/c2_kernel_trace.csv $O/kt_c2_bench.json "$C --workload c2" \
  $O/kt_c3_L5.3/c3_L5.3_kernel_trace.csv $O/kt_c3_L5.3_bench.json "$C --clip-blocks 5.3" \
  $O/kt_i16r/i16r_kernel_trace.csv $O/kt_i16r_bench.json "$C --workload i16r" \
  $O/kt_c3_K256/c3_K256_kernel_trace.csv $O/kt_c3_K256_bench.json "$C --blocks 256 (grouped order)" > profiles/${RN}_kernel_summary.txt
for n in c3 c2 c4 c3_L5.3; do
  python tools/timeline.py $O/kt_$n/${n}_kernel_trace.csv 16 > profiles/${RN}_timeline_$n.txt
done
KH=$(kof $O/kt_c3_bench.json)
python tools/pmc_traffic.py $O/pmc_c3 c3 $KH 4096 mix_kernel > /dev/null
python tools/pmc_traffic.py $O/pmc_c4 c4 $KH 4096 mix_kernel > /dev/null
python tools/pmc_traffic.py $O/pmc_c3_L5.3 c3 $KH 4096 mix_kernel _L5.3 > /dev/null
python tools/pmc_traffic.py $O/pmc_i16r i16r $KH 4096 mix_kernel > /dev/null
python tools/pmc_traffic.py $O/pmc_c3_K256 c3 256 4096 mix_kernel > /dev/null
python - $TAG $RN $KH <<'PY'
import json, sys
tag, rn, K = sys.argv[1], sys.argv[2], sys.argv[3]
p = 'profiles/pmc_traffic.json'; d = json.load(open(p))
m = {f'c3_K{K}_N4096': ('c3', K), f'c3_K{K}_N4096_L5.3': ('c3_L5.3', K), f'c4_K{K}_N4096': ('c4', K), f'i16r_K{K}_N4096': ('i16r', K),
     'c3_K256_N4096': ('c3', '256')}
for k, (n, kk) in m.items():
    d[k]['source'] = f'profiles/{rn}_{n}_K{kk}_mix_pmc.txt (tools/profile_round.sh; raw rocprofv3 --pmc CSVs in gpurun_out/{tag}/, scratch)'
json.dump(d, open(p, 'w'), indent=1, sort_keys=True); open(p, 'a').write('\n')
for k, v in d.items(): print(k, v['hbm_bytes_per_launch'])
# the kernel a bench line names is the kernel rocprofv3 saw in the same run
for n, kk in m.values():
    name = json.load(open(f'profiles/{rn}_{n}_K{kk}_bench_line_of_profiled_run.json'))['roofline']['kernel']
    stats = open(f'profiles/{rn}_{n}_K{kk}_kernel_stats_rocprofv3.csv').read()
    assert name in stats, (n, name)
    print(n, kk, 'bench line and rocprofv3 agree on', name)
PY
cat profiles/${RN}_kernel_summary.txt
