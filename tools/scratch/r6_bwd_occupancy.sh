#!/bin/bash
# round 6: the backward kernel's passes per launch x waves per SIMD at nemb <= 16 (variants of fused_bwd_mfma_e16.hip + its dispatcher)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
for v in "" bwd_p2_b2 bwd_p2_b3 bwd_p2_b4 bwd_p1_b4 bwd_p1_b3; do
  if [ -n "$v" ]; then export ARMNET_HIP_LIB=$ROOT/arm-net_amd/lib/exp/libarmnet_$v.so; else unset ARMNET_HIP_LIB; fi
  echo "## ${v:-product (4 passes, 2 waves per SIMD)}"
  python tools/bwd_bench.py --steps 30 2>&1 | grep -v amdgpu.ids
  python tools/bwd_bench.py --steps 30 --alpha 1.7 2>&1 | grep -v amdgpu.ids
  python tools/bwd_bench.py --steps 30 --O 128 2>&1 | grep -v amdgpu.ids
  python tools/bwd_bench.py --steps 30 --F 10 --E 10 --O 256 2>&1 | grep -v amdgpu.ids
done
