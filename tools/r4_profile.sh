#!/bin/bash
# round 4 (GPU box, through gpurun): the evidence DESIGN.md's table and bench.py's line quote, for the FINAL kernel sources
#   1. rocprofv3 kernel trace + PMC passes of the headline workload, fresh regime; timings + SQ pass for the stress regime
#   2. tools/ubench/gather_stream (the access pattern's own ceiling); pmc_traffic.json / access_pattern_ceiling.json
#   3. the default bench.py line (twice: a freshly leased box and right after)
#   4. tools/sweep.sh (measurement matrix), kbench on the nemb > 64 shapes, the row-sharded path on one rank, routing alone
#   5. tools/parity_margin.py (worst absolute errors per fixture family)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
G=$ROOT/gpurun_out
cd "$ROOT"
python bench.py --steps 20 --warmup 5 > $G/r4_bench_n1_first.json 2> $G/r4_bench_n1_first.err
python bench.py --steps 20 --warmup 5 > $G/r4_bench_n1.json 2> $G/r4_bench_n1.err
PMC_EXTRA="--settle-ms 0" bash tools/profile.sh r4_prof_fresh -- python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-other-alphas --regime fresh --in-flight 1
python tools/prof_summary.py gpurun_out/r4_prof_fresh fused > $G/r4_prof_fresh_summary.txt 2>&1
python tools/prof_summary.py gpurun_out/r4_prof_fresh mlp_head >> $G/r4_prof_fresh_summary.txt 2>&1
PROFILE_LIGHT=1 PMC_EXTRA="--settle-ms 0" bash tools/profile.sh r4_prof_stress -- python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-other-alphas --regime stress --in-flight 1
python tools/prof_summary.py gpurun_out/r4_prof_stress fused > $G/r4_prof_stress_summary.txt 2>&1
cd "$ROOT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench/gather_stream tools/ubench/gather_stream.hip && tools/ubench/gather_stream 4 > $G/r4_gather_stream.txt 2>&1
python tools/make_profile_json.py gpurun_out/r4_prof_fresh gpurun_out/r4_gather_stream.txt r04 > $G/r4_make_profile_json.log 2>&1
cp profiles/pmc_traffic.json $G/r4_pmc_traffic.json; cp profiles/access_pattern_ceiling.json $G/r4_access_pattern_ceiling.json
python bench.py --steps 20 --warmup 5 > $G/r4_bench_n1_with_traffic.json 2> $G/r4_bench_n1_with_traffic.err
{
echo "# tools/sweep.sh, round 4 (bench.py --steps 30 --warmup 5 --no-cpu-baseline + the flags of each line), 1 x MI355X, value = median of windows"
bash tools/sweep.sh
echo
echo "# the row-sharded lookup path on ONE rank, headline shape (bench.py --shard rows ...): whole-shard exchange | request lists de-duplicated | request lists"
P='import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print("%-60s %7.1f us/step %7.1f Msamp/s  full fwd %7.1f Msamp/s" % (sys.argv[1], d["ms_per_step"]*1e3, d["value"]/1e6, d["full_forward"]["value"]/1e6))'
for v in "--shard rows" "--shard rows --whole-shard off" "--shard rows --whole-shard off --dedup off" "--shard rows --nemb 64 --nfeat 100000000"; do
  python bench.py $v --steps 30 --warmup 5 --no-cpu-baseline --no-other-alphas --regime fresh 2>/dev/null | python -c "$P" "$v"
done
echo
echo "# tools/route_bench.py: routing of the fixed-capacity protocol alone"
python tools/route_bench.py 2>&1 | grep -v amdgpu.ids
echo
echo "# tools/kbench.py: nemb > 64 on the matrix cores (flags 0) against the shape-agnostic kernel (flags 0x4), and the run.sh / BASELINE block shapes"
for cfg in "10 100 10 1.7" "39 96 32 2.0" "22 72 32 2.0" "39 128 32 2.0" "10 128 128 1.7"; do
  set -- $cfg
  python tools/kbench.py --F $1 --E $2 --O $3 --alpha $4 --flags 0 0x4 2>&1 | grep -v amdgpu.ids
done
for cfg in "39 16 32 65536" "39 16 128 65536" "39 64 32 65536" "22 32 128 131072" "10 10 256 65536" "3 10 128 65536" "22 10 128 65536" "22 10 64 65536" "39 10 128 65536" "39 10 256 65536" "43 10 32 65536" "43 10 512 65536"; do
  set -- $cfg
  python tools/kbench.py --F $1 --E $2 --O $3 --B $4 2>&1 | tail -1
  python tools/bwd_bench.py --F $1 --E $2 --O $3 --B $4 2>&1 | tail -1
done
python tools/kbench.py --alpha 2.5 2>&1 | tail -1
python tools/kbench.py --alpha 2.5 --F 3 --E 10 --O 128 2>&1 | tail -1
echo
echo "# tools/mlp_bench.py"
python tools/mlp_bench.py 2>&1 | grep -v amdgpu.ids
echo
echo "# tools/bwd_bench.py (backward of the block) at the README's nemb = 100 shape and the headline shape"
python tools/bwd_bench.py --F 10 --E 100 --O 10 --alpha 1.7 2>&1 | grep -v amdgpu.ids
python tools/bwd_bench.py 2>&1 | grep -v amdgpu.ids
} > $G/r4_bench_variants_1gpu.txt 2>&1
python tools/parity_margin.py > $G/r4_parity_margin.txt 2>&1
tail -c 400 $G/r4_bench_n1.json; echo; tail -30 $G/r4_bench_variants_1gpu.txt; cat $G/r4_parity_margin.txt
