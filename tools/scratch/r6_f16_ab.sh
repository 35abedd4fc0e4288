#!/bin/bash
# round 6: the ARM block with both contractions as fp16 x 2 splits (product default from 64 neurons per launch) against the fp32 MFMAs
# (flag 0x10 = ARMNET_F_FP32_CONTRACTIONS), same library, same process
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
kb() { python tools/kbench.py --steps 100 "$@" --flags 0 0x10 2>&1 | grep -v amdgpu.ids; }
for O in 64 128 256; do
  kb --O $O --regime fresh; kb --O $O --regime stress; kb --O $O --regime stress --alpha 1.7; kb --O $O --regime fresh --alpha 1.5; kb --O $O --regime fresh --alpha 1.0
done
kb --F 39 --E 10 --O 256 --regime fresh
kb --F 10 --E 10 --O 256 --regime fresh; kb --F 10 --E 10 --O 256 --regime stress --alpha 1.7
kb --F 10 --E 10 --O 64 --regime fresh
kb --F 22 --E 16 --O 128 --regime fresh; kb --F 22 --E 16 --O 128 --regime stress
kb --F 22 --E 10 --O 512 --regime fresh
kb --F 30 --E 16 --O 128 --regime fresh; kb --F 43 --E 16 --O 128 --regime fresh; kb --F 43 --E 10 --O 256 --regime stress
kb --F 3 --E 10 --O 128 --regime fresh; kb --F 7 --E 16 --O 64 --regime fresh
