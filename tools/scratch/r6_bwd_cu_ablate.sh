#!/bin/bash
# round 6: the block's backward kernel on 64 / 128 / 256 CUs with its ablation switches (-DARMNET_DEV_FLAGS build):
# 0x400 no d_table atomics, 0x200 ids folded into 1 024 rows, 0x2000 only the first 16-neuron pass, 0x4000 no block flush
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
export ARMNET_HIP_LIB=$ROOT/arm-net_amd/lib/exp/libarmnet_devflags.so
for cus in 64 128 256; do
  python tools/bwd_bench.py --steps 30 --cus $cus --flags 0 0x400 0x200 0x600 0x2000 0x2600 0x4000 2>&1 | grep -v amdgpu.ids
done
python tools/bwd_bench.py --steps 30 --alpha 1.7 --flags 0 0x600 2>&1 | grep -v amdgpu.ids
