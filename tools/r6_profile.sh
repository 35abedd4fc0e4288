#!/bin/bash
# round 6 (GPU box, through gpurun): the evidence DESIGN.md's table and bench.py's line quote, for the FINAL kernel sources
#   1. the default bench.py line, twice (a freshly leased box and right after)
#   2. rocprofv3 kernel trace + PMC passes of the headline workload, fresh regime; timings + SQ pass for the stress regime
#   3. tools/ubench/gather_stream (the access pattern's own ceiling); pmc_traffic.json / access_pattern_ceiling.json; the bench
#      line again with the traffic of this kernel-source hash
#   4. tools/sweep.sh (measurement matrix), the row-sharded path on one rank (with and without hot rows), routing alone
#   5. tools/parity_margin.py (worst absolute errors per fixture family)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
G=$ROOT/gpurun_out
mkdir -p $G
cd "$ROOT"
python bench.py --steps 20 --warmup 5 > $G/r6_bench_n1_first.json 2> $G/r6_bench_n1_first.err
python bench.py --steps 20 --warmup 5 > $G/r6_bench_n1.json 2> $G/r6_bench_n1.err
PMC_EXTRA="--settle-ms 0" bash tools/profile.sh r6_prof_fresh -- python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-other-alphas --regime fresh --in-flight 1
python tools/prof_summary.py gpurun_out/r6_prof_fresh fused > $G/r6_prof_fresh_summary.txt 2>&1
python tools/prof_summary.py gpurun_out/r6_prof_fresh mlp_head >> $G/r6_prof_fresh_summary.txt 2>&1
PROFILE_LIGHT=1 PMC_EXTRA="--settle-ms 0" bash tools/profile.sh r6_prof_stress -- python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-other-alphas --regime stress --in-flight 1
python tools/prof_summary.py gpurun_out/r6_prof_stress fused > $G/r6_prof_stress_summary.txt 2>&1
cd "$ROOT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench/gather_stream tools/ubench/gather_stream.hip && tools/ubench/gather_stream 4 > $G/r6_gather_stream.txt 2>&1
python tools/make_profile_json.py gpurun_out/r6_prof_fresh gpurun_out/r6_gather_stream.txt r06 > $G/r6_make_profile_json.log 2>&1
cp profiles/pmc_traffic.json $G/r6_pmc_traffic.json; cp profiles/access_pattern_ceiling.json $G/r6_access_pattern_ceiling.json
python bench.py --steps 20 --warmup 5 > $G/r6_bench_n1_with_traffic.json 2> $G/r6_bench_n1_with_traffic.err
{
echo "# tools/sweep.sh, round 6 (bench.py --steps 30 --warmup 5 --no-cpu-baseline + the flags of each line), 1 x MI355X, value = median of windows"
bash tools/sweep.sh
echo
echo "# the row-sharded lookup path on ONE rank (bench.py --shard rows ...): headline shape and configs[3]; uniform and skewed ids; hot rows"
P='import sys,json
d=json.loads(sys.stdin.readlines()[-1]); h=d.get("hot_rows",{}); print("%-78s %7.1f us/step %7.1f Msamp/s  full fwd %7.1f Msamp/s  path %s  overflow %s  ingress@8 %s" % (sys.argv[1], d["ms_per_step"]*1e3, d["value"]/1e6, d["full_forward"]["value"]/1e6, d.get("row_sharded_path"), d.get("row_sharded_overflow"), json.dumps(h.get("ingress_bytes_per_rank_per_step_at_8_ranks"))))'
for v in "--shard rows" "--shard rows --whole-shard off" "--shard rows --whole-shard off --dedup off" \
         "--shard rows --ids zipf --whole-shard off" "--shard rows --ids zipf --whole-shard off --hot-rows 65536" \
         "--shard rows --ids zipf --whole-shard off --dedup off" "--shard rows --ids zipf --whole-shard off --dedup off --hot-rows 65536" \
         "--shard rows --nemb 64 --nfeat 100000000" "--shard rows --ids zipf --nemb 64 --nfeat 100000000" \
         "--shard rows --ids zipf --nemb 64 --nfeat 100000000 --hot-rows 65536" "--shard rows --ids zipf --nemb 64 --nfeat 100000000 --hot-rows 1048576"; do
  python bench.py $v --steps 30 --warmup 5 --no-cpu-baseline --no-other-alphas --regime fresh 2>/dev/null | python -c "$P" "$v"
done
echo
echo "# tools/route_bench.py: routing of the fixed-capacity protocol alone"
python tools/route_bench.py 2>&1 | grep -v amdgpu.ids
echo
echo "# tools/kbench.py: the run.sh / BASELINE block shapes, forward (and tools/bwd_bench.py, backward)"
for cfg in "39 16 32 65536" "39 16 128 65536" "39 64 32 65536" "22 32 128 131072" "10 10 256 65536" "3 10 128 65536" "22 10 128 65536" "22 10 64 65536" "39 10 128 65536" "39 10 256 65536" "43 10 32 65536" "43 10 512 65536" "10 100 10 65536"; do
  set -- $cfg
  python tools/kbench.py --F $1 --E $2 --O $3 --B $4 2>&1 | tail -1
  python tools/bwd_bench.py --F $1 --E $2 --O $3 --B $4 2>&1 | tail -1
done
python tools/kbench.py --alpha 2.5 2>&1 | tail -1
echo
echo "# tools/mlp_bench.py"
python tools/mlp_bench.py 2>&1 | grep -v amdgpu.ids
} > $G/r6_bench_variants_1gpu.txt 2>&1
python tools/parity_margin.py > $G/r6_parity_margin.txt 2>&1
tail -c 600 $G/r6_bench_n1.json; echo; tail -40 $G/r6_bench_variants_1gpu.txt; tail -22 $G/r6_parity_margin.txt
