#!/bin/bash
# Turn the output of tools/profile_r02.sh <tag> (gpurun_out/<tag>/, scratch) into the committed files under profiles/.
# usage (in the repo root): tools/refresh_profiles.sh <tag>
set -e
TAG=${1:?tag}
O=gpurun_out/$TAG
for f in $O/bench_*.json; do W=$(basename $f .json); W=${W#bench_}; grep "^{" $f | tail -1 > profiles/r02_bench_$W.json; done
cp $O/pytest_gpu.log profiles/r02_pytest_gpu.log
for n in c3 c4 c3_L5.3 i16r; do
  cp $O/kt_$n/${n}_kernel_stats.csv profiles/r02_${n}_K256_kernel_stats_rocprofv3.csv
  grep "^{" $O/kt_${n}_bench.json | tail -1 > profiles/r02_${n}_K256_bench_line_of_profiled_run.json
  python tools/pmc_summary.py $O/pmc_$n mix_kernel > profiles/r02_${n}_K256_mix_pmc.txt
done
cp $O/kt_c3_noov/c3_kernel_stats.csv profiles/r02_c3_K256_kernel_stats_rocprofv3_no_overlap.csv
C="python bench.py --no-cpu-baseline --no-configs --latency-blocks 0"
python tools/kernel_summary.py $O/kt_c3/c3_kernel_trace.csv $O/kt_c3_bench.json "$C" \
  $O/kt_c3_noov/c3_kernel_trace.csv $O/kt_c3_noov_bench.json "WBX_OVERLAP=0 (plan on the main stream), same command" \
  $O/kt_c4/c4_kernel_trace.csv $O/kt_c4_bench.json "$C --workload c4" \
  $O/kt_c3_L5.3/c3_L5.3_kernel_trace.csv $O/kt_c3_L5.3_bench.json "$C --clip-blocks 5.3" \
  $O/kt_i16r/i16r_kernel_trace.csv $O/kt_i16r_bench.json "$C --workload i16r" > profiles/r02_kernel_summary.txt
python tools/pmc_traffic.py $O/pmc_c3 c3 256 4096 mix_kernel > /dev/null
python tools/pmc_traffic.py $O/pmc_c4 c4 256 4096 mix_kernel > /dev/null
python tools/pmc_traffic.py $O/pmc_c3_L5.3 c3 256 4096 mix_kernel _L5.3 > /dev/null
python tools/pmc_traffic.py $O/pmc_i16r i16r 256 4096 mix_kernel > /dev/null
python - $TAG <<'PY'
import json, sys
tag = sys.argv[1]
p = 'profiles/pmc_traffic.json'; d = json.load(open(p))
m = {'c3_K256_N4096': 'c3', 'c3_K256_N4096_L5.3': 'c3_L5.3', 'c4_K256_N4096': 'c4', 'i16r_K256_N4096': 'i16r'}
for k, n in m.items():
    d[k]['source'] = f'profiles/r02_{n}_K256_mix_pmc.txt (tools/profile_r02.sh; raw rocprofv3 --pmc CSVs in gpurun_out/{tag}/pmc_{n}, scratch)'
json.dump(d, open(p, 'w'), indent=1, sort_keys=True); open(p, 'a').write('\n')
for k, v in d.items(): print(k, v['hbm_bytes_per_launch'])
# the kernel a bench line names is the kernel rocprofv3 saw in the same run
for n in m.values():
    name = json.load(open(f'profiles/r02_{n}_K256_bench_line_of_profiled_run.json'))['roofline']['kernel']
    stats = open(f'profiles/r02_{n}_K256_kernel_stats_rocprofv3.csv').read()
    assert name in stats, (n, name)
    print(n, 'bench line and rocprofv3 agree on', name)
PY
cat profiles/r02_kernel_summary.txt
