#!/bin/bash
# round 6, after the last change to the forward kernel's sources: the full GPU suite, then the counter passes + bench line again
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; G=$ROOT/gpurun_out; mkdir -p $G; cd "$ROOT"
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $G/r6_gpu_suite_final.log
PMC_EXTRA="--settle-ms 0" bash tools/profile.sh r6_prof_fresh -- python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-other-alphas --regime fresh --in-flight 1
python tools/prof_summary.py gpurun_out/r6_prof_fresh fused > $G/r6_prof_fresh_summary.txt 2>&1
python tools/prof_summary.py gpurun_out/r6_prof_fresh mlp_head >> $G/r6_prof_fresh_summary.txt 2>&1
cd "$ROOT"
tools/ubench/gather_stream 4 > $G/r6_gather_stream.txt 2>&1
python tools/make_profile_json.py gpurun_out/r6_prof_fresh gpurun_out/r6_gather_stream.txt r06 > $G/r6_make_profile_json.log 2>&1
cp profiles/pmc_traffic.json $G/r6_pmc_traffic.json; cp profiles/access_pattern_ceiling.json $G/r6_access_pattern_ceiling.json
python bench.py --steps 20 --warmup 5 > $G/r6_bench_n1_with_traffic.json 2> $G/r6_bench_n1_with_traffic.err
cat $G/r6_gpu_suite_final.log; tail -c 300 $G/r6_bench_n1_with_traffic.json
