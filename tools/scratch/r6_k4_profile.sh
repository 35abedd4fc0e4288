#!/bin/bash
# round 6: rocprofv3 timings + counters of configs[2]'s block with both contractions as fp16 x 2 splits, and with the fp32 MFMAs (kernel flag)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
PMC_EXTRA="--settle-ms 0" bash tools/profile.sh r6_prof_k4 -- python "$ROOT/bench.py" --nhead 4 --steps 20 --warmup 5 --no-cpu-baseline --no-other-alphas --regime fresh --in-flight 1
cd "$ROOT"; python tools/prof_summary.py gpurun_out/r6_prof_k4 fused > gpurun_out/r6_prof_k4_summary.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r6_prof_k4_kb -- python $ROOT/tools/kbench.py --steps 200 --O 128 --flags 0 0x10 > $ROOT/gpurun_out/r6_prof_k4_kb.log 2>&1
cd "$ROOT"; python tools/prof_summary.py gpurun_out/r6_prof_k4_kb fused >> gpurun_out/r6_prof_k4_summary.txt 2>&1
cat gpurun_out/r6_prof_k4_summary.txt | cut -c1-200
