#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
P='import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print("%-60s %7.1f us/step %7.1f Msamp/s  full fwd %7.1f Msamp/s" % (sys.argv[1], d["ms_per_step"]*1e3, d["value"]/1e6, d["full_forward"]["value"]/1e6))'
for v in "--shard rows" "--shard rows --whole-shard off" "--shard rows --whole-shard off --dedup off" "--shard rows --nemb 64 --nfeat 100000000"; do
  python bench.py $v --steps 30 --warmup 5 --no-cpu-baseline --no-other-alphas --regime fresh 2>/dev/null | python -c "$P" "$v"
done
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_bench_contract.py -x -q -m gpu 2>&1 | tail -4
