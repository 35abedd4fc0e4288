#!/bin/bash
# round 6: the headline kernel's ablation switches (-DARMNET_DEV_FLAGS build of fused_mfma_e16.hip) on 64 and on 256 CUs
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
export ARMNET_HIP_LIB=$ROOT/arm-net_amd/lib/exp/libarmnet_devflags.so
for cus in 64 128 256; do
  python tools/kbench.py --steps 100 --cus $cus --flags 0 0x800 0x100 0x400 0x200 0x600 0xe00 0xf00 0x900
  python tools/kbench.py --steps 100 --cus $cus --regime stress --flags 0 0x800 0x100
done
