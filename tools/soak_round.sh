#!/bin/bash
# Soak runs of a round on one GPU box (one gpurun call): fresh seeds of the parity generators and of the differential against
# the live reference executable (oracle/_ref/wbref_engine travels with the tree).  -> gpurun_out/<tag>/soak.txt
# usage (on the GPU box): tools/soak_round.sh [tag] [seeds per script kind] [seed base of the live-reference scripts] [seed base of the parity generators]
set -u
TAG=${1:-r05}
N=${2:-400}
LB=${3:-20000}
PB=${4:-300000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
{
  echo "# tools/soak_round.sh $TAG $N   ($(date -u +%FT%TZ))"
  echo "== the product against the live reference executable: $N seeds per script kind from $LB (callback for even seeds, batch renders for odd)"
  WBX_REFSEQ_GPU_SEEDS=$N WBX_REFSEQ_GPU_FROM=$LB timeout 1500 python -m pytest tests/test_gpu_refseq.py -m gpu -q -k live 2>&1 | tail -3
  echo "== parity generators on fresh seeds from $PB (product against the oracle)"
  E=""; for k in "" 2 3 4 5 6 7 8 9 10; do E="$E WBX_FUZZ${k}_FROM=$PB WBX_FUZZ${k}_TO=$((PB + 300))"; done
  env $E timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 4 2>&1 | tail -3
  echo "== edit scripts / destroyed clips / wild sessions / segmented plans (their own default seeds: the final tree)"
  timeout 1200 python -m pytest tests/test_destroyed_clip.py tests/test_wild.py tests/test_gpu_segments.py tests/test_gpu_ragged.py -m gpu -q 2>&1 | tail -3
} > $O/soak.txt 2>&1
cat $O/soak.txt
