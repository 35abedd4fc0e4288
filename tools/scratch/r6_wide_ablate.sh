#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
export ARMNET_HIP_LIB=$ROOT/arm-net_amd/lib/exp/libarmnet_devflags.so
for r in fresh stress; do
python tools/kbench.py --steps 50 --O 128 --regime $r --flags 0 0x800 0x100 0x900 0x400 0x600 0xe00 0xf00 2>&1 | grep -v amdgpu.ids
done
python tools/kbench.py --steps 50 --O 128 --regime stress --alpha 1.7 --flags 0 0x800 0x100 2>&1 | grep -v amdgpu.ids
python tools/kbench.py --steps 50 --O 64 --regime fresh --flags 0 0x800 2>&1 | grep -v amdgpu.ids
