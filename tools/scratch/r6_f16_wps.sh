#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
for v in "" f16w5 f16w3; do
  if [ -n "$v" ]; then export ARMNET_HIP_LIB=$ROOT/arm-net_amd/lib/exp/libarmnet_$v.so; else unset ARMNET_HIP_LIB; fi
  echo "## ${v:-product (4 waves per SIMD)}"
  for args in "--O 128 --regime fresh" "--O 128 --regime stress" "--O 128 --regime stress --alpha 1.7" "--O 256 --regime fresh" "--O 64 --regime fresh" "--O 40 --regime stress" "--F 43 --E 10 --O 256 --regime stress"; do
    python tools/kbench.py --steps 100 $args 2>&1 | grep -v amdgpu.ids
  done
done
