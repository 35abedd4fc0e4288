#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
export ARMNET_HIP_LIB=$ROOT/arm-net_amd/lib/exp/libarmnet_f16min33.so
kb() { python tools/kbench.py --steps 100 "$@" --flags 0 0x10 2>&1 | grep -v amdgpu.ids; }
for O in 40 48 56; do kb --O $O --regime fresh; kb --O $O --regime stress; kb --O $O --regime stress --alpha 1.7; done
kb --O 48 --F 43 --regime fresh; kb --O 48 --F 30 --regime fresh
