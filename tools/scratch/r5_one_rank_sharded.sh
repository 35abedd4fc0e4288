cd $GRAFT_REPO_ROOT
P='import sys,json
d=json.loads(sys.stdin.readlines()[-1]); h=d.get("hot_rows",{}); print("%-78s %7.1f us/step %7.1f Msamp/s  full fwd %7.1f Msamp/s  path %s  overflow %s  ingress@8 %s" % (sys.argv[1], d["ms_per_step"]*1e3, d["value"]/1e6, d["full_forward"]["value"]/1e6, d.get("row_sharded_path"), d.get("row_sharded_overflow"), json.dumps(h.get("ingress_bytes_per_rank_per_step_at_8_ranks"))))'
{
echo "# the row-sharded lookup path on ONE rank (bench.py --shard rows ...), re-run after slots stopped being clamped to their first-step estimate: headline shape and configs[3]; uniform and skewed ids; hot rows"
for v in "--shard rows" "--shard rows --whole-shard off" "--shard rows --whole-shard off --dedup off" \
         "--shard rows --ids zipf --whole-shard off" "--shard rows --ids zipf --whole-shard off --hot-rows 65536" \
         "--shard rows --ids zipf --whole-shard off --dedup off" "--shard rows --ids zipf --whole-shard off --dedup off --hot-rows 65536" \
         "--shard rows --nemb 64 --nfeat 100000000" "--shard rows --ids zipf --nemb 64 --nfeat 100000000" \
         "--shard rows --ids zipf --nemb 64 --nfeat 100000000 --hot-rows 65536" "--shard rows --ids zipf --nemb 64 --nfeat 100000000 --hot-rows 1048576"; do
  python bench.py $v --steps 30 --warmup 5 --no-cpu-baseline --no-other-alphas --regime fresh 2>/dev/null | python -c "$P" "$v"
done
} > gpurun_out/r5_one_rank_sharded.txt 2>&1
cat gpurun_out/r5_one_rank_sharded.txt
timeout 300 python -m pytest tests/test_bench_contract.py tests/test_sharded_gpu.py -m gpu -q 2>&1 | tail -2
