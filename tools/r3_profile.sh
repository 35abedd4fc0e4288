#!/bin/bash
# round 3 (GPU box, through gpurun): the evidence bench.py's line quotes, for the FINAL kernel sources
#   1. rocprofv3 kernel trace + PMC passes of the headline workload, fresh regime (profiles/r03_bench_n1_*)
#   2. timings + SQ pass 1 for the stress regime (SQ_INSTS_VALU per launch, both regimes)
#   3. tools/ubench/gather_stream (the access pattern's own ceiling)
#   4. the default bench.py line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
PMC_EXTRA="--settle-ms 0" bash tools/profile.sh r3_prof_fresh -- python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-other-alphas --regime fresh --in-flight 1
python tools/prof_summary.py gpurun_out/r3_prof_fresh fused > gpurun_out/r3_prof_fresh_summary.txt 2>&1
python tools/prof_summary.py gpurun_out/r3_prof_fresh mlp_head >> gpurun_out/r3_prof_fresh_summary.txt 2>&1
PROFILE_LIGHT=1 PMC_EXTRA="--settle-ms 0" bash tools/profile.sh r3_prof_stress -- python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-other-alphas --regime stress --in-flight 1
python tools/prof_summary.py gpurun_out/r3_prof_stress fused > gpurun_out/r3_prof_stress_summary.txt 2>&1
cd "$ROOT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench/gather_stream tools/ubench/gather_stream.hip && tools/ubench/gather_stream 4 > gpurun_out/r3_gather_stream.txt 2>&1
python tools/make_profile_json.py gpurun_out/r3_prof_fresh gpurun_out/r3_gather_stream.txt r03 > gpurun_out/r3_make_profile_json.log 2>&1
cp profiles/pmc_traffic.json gpurun_out/r3_pmc_traffic.json; cp profiles/access_pattern_ceiling.json gpurun_out/r3_access_pattern_ceiling.json
python bench.py > gpurun_out/r3_bench_n1.json 2> gpurun_out/r3_bench_n1.err
tail -c 600 gpurun_out/r3_bench_n1.json
