#!/bin/bash
# callback latency (wbx_engine_process) at the low-latency block sizes: one launch through the 256-lane instance (default)
# against the three launches of the block's own shape (WBX_CB_ANY=0).   usage (GPU box): tools/exp_cb_blocks.sh [tag]
set -u
O=gpurun_out/${1:-r05_cb}; mkdir -p $O
for F in 128 256 480 512 64; do
  for ANY in 1 0; do
    echo "== F=$F WBX_CB_ANY=$ANY" | tee -a $O/callback_block_sizes.txt
    WBX_CB_ANY=$ANY timeout 300 python bench.py --block-frames $F --no-cpu-baseline --no-configs --no-verify --steps 2 --warmup 1 --ramp-steps 2 --latency-blocks 800 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d['latency_mode']))" | tee -a $O/callback_block_sizes.txt
  done
done
