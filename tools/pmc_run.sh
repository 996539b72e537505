#!/bin/bash
# Collect rocprofv3 PMC counters for the mix path, one counter group per pass (separate runs, no trace
# domains other than kernel-trace).  usage: tools/pmc_run.sh <workload> <outdir> [extra bench args]
set -u
W=$1; OUT=$2; shift 2
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
cd /tmp
pass() {
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "mix_kernel" --pmc "$@" --output-format csv -d "$OUT/$name" -o $W -- \
    python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --latency-blocks 0 $EXTRA > "$OUT/$name.log" 2>&1
}
EXTRA="$*"
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA
pass write WRITE_SIZE TCC_HIT TCC_MISS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32
pass tcp TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_BRANCH
