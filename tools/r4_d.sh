#!/bin/bash
# round 4, GPU call D: bisection with known-sign steps — bit identity and time; routing after the emit rewrite
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4d
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_headline.py -x -q -m gpu -k "bisect or alpha25 or a2.5 or faithful or route_fixed or routing" 2>&1 | tail -12
{
python tools/kbench.py --alpha 2.5 --flags 0 0x8
python tools/kbench.py --alpha 2.5 --regime stress --flags 0 0x8
python tools/kbench.py --alpha 3.0 --flags 0 0x8
python tools/kbench.py --alpha 2.5 --F 3 --E 10 --O 128 --flags 0 0x8
python tools/kbench.py --alpha 2.5 --F 22 --E 10 --O 64 --flags 0 0x8
python tools/kbench.py --alpha 2.0 --flags 0x2 0xa
python tools/kbench.py --alpha 1.7 --regime stress --flags 0x2 0xa
} 2>&1 | grep -v amdgpu.ids | tee "$OUT/kbench_bisect.txt"
timeout 600 python tools/route_bench.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/route_bench.txt"
