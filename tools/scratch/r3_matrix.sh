#!/bin/bash
# round 3 (GPU box): the measurement matrix + where-the-time-goes tables quoted in DESIGN.md
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
{ echo "# tools/fwd_parts.py"; python tools/fwd_parts.py 2>/dev/null; echo; echo "# tools/fwd_parts.py --nhead 4 (config 3)"; python tools/fwd_parts.py --nhead 4 2>/dev/null; } > gpurun_out/r3_fwd_parts.txt
{ for cfg in "2.0 fresh 0 32" "2.0 stress 0 32" "2.0 fresh 0 128" "1.7 stress 0 32"; do set -- $cfg; ARMNET_HIP_LIB=$ROOT/arm-net_amd/lib/libarmnet_phase.so python tools/phase_timing.py $1 $2 $3 $4 2>/dev/null; echo; done; } > gpurun_out/r3_phase_timing.txt
bash tools/sweep.sh > gpurun_out/r3_sweep.txt 2>&1
{ for cfg in "39 16 32 65536" "10 10 32 65536" "3 10 32 65536" "22 10 32 65536" "39 16 128 65536" "39 64 32 65536" "22 32 128 131072" "10 10 256 65536" "3 10 128 65536" "22 10 128 65536" "22 10 64 65536" "39 10 128 65536" "39 10 256 65536" "43 10 32 65536" "43 10 512 65536"; do set -- $cfg; python tools/kbench.py --F $1 --E $2 --O $3 --B $4 2>&1 | tail -1; done; python tools/kbench.py --F 3 --E 10 --O 128 --alpha 2.5 2>&1 | tail -1; } > gpurun_out/r3_block_shapes.txt
python tools/mlp_bench.py > gpurun_out/r3_mlp_bench.txt 2>&1
tail -3 gpurun_out/r3_sweep.txt
