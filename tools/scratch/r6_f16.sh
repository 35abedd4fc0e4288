#!/bin/bash
# round 6: both contractions as fp16 x 2 splits (F16) against the fp32 MFMAs (flag 0x10) in the same library; parity tests on the F16 path
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
export ARMNET_HIP_LIB=$ROOT/arm-net_amd/lib/exp/libarmnet_f16.so
for O in 32 64 128; do
  for r in fresh stress; do
    python tools/kbench.py --steps 100 --O $O --regime $r --flags 0 0x10 2>&1 | grep -v amdgpu.ids
  done
  python tools/kbench.py --steps 100 --O $O --regime stress --alpha 1.7 --flags 0 0x10 2>&1 | grep -v amdgpu.ids
  python tools/kbench.py --steps 100 --O $O --regime fresh --alpha 1.5 --flags 0 0x10 2>&1 | grep -v amdgpu.ids
done
python tools/kbench.py --steps 100 --O 256 --F 39 --E 10 --regime fresh --flags 0 0x10 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_headline.py -m gpu -x -q 2>&1 | tail -15
