#!/bin/bash
# tools/pmc_short.sh — issue counters of the short-block instances on the UNCUT c3 session: the packed instance (default) against
# the one-block-per-workgroup instance a session cut into clips takes (WBX_FORCE_CUT=1).  128- and 256-frame stereo blocks.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmc_short; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for F in 128 256; do for V in packed onewave; do
  E=""; [ $V = onewave ] && E="WBX_FORCE_CUT=1"
  env $E timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "mix_kernel" --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d $O/F${F}_$V -o p -- python $R/bench.py --block-frames $F --blocks 1024 --steps 3 --warmup 1 --ramp-steps 3 --no-cpu-baseline --no-configs --no-verify --latency-blocks 0 > $O/F${F}_$V.log 2>&1
  echo "== F=$F $V"; python $R/tools/pmc_summary.py $O/F${F}_$V mix_kernel | grep -v "^#"
done; done
