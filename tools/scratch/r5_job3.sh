#!/bin/bash
# round 5, GPU call 3: full GPU suite on the LIN build, A/B of the first-order solver finish (libarmnet_nolin.so = -DARMNET_NO_LIN),
# parity margins, the row-sharded path on one rank with and without hot rows under skewed ids
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
G=$ROOT/gpurun_out
mkdir -p $G
cd "$ROOT"
{
echo "# tools/kbench.py, B = 65536, 39 fields, nemb 16, 32 neurons: product (first-order finish) | -DARMNET_NO_LIN"
for alpha in 1.7 1.5 1.3 1.9 2.0; do
  for regime in fresh stress; do
    for lib in lib/libarmnet_hip.so lib/exp/libarmnet_nolin.so; do
      ARMNET_HIP_LIB=$ROOT/arm-net_amd/$lib python tools/kbench.py --alpha $alpha --regime $regime --steps 100 2>&1 | tail -1 | sed "s|^|$lib: |"
    done
  done
done
echo "# 128 neurons"
for alpha in 1.7 1.5; do
  for lib in lib/libarmnet_hip.so lib/exp/libarmnet_nolin.so; do
    ARMNET_HIP_LIB=$ROOT/arm-net_amd/$lib python tools/kbench.py --alpha $alpha --regime stress --O 128 --steps 60 2>&1 | tail -1 | sed "s|^|$lib: |"
  done
done
} > $G/r5_lin_ab.txt 2>&1
cat $G/r5_lin_ab.txt
timeout 600 python tools/parity_margin.py > $G/r5_parity_margin.txt 2>&1; tail -25 $G/r5_parity_margin.txt
{
P='import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print("%-70s %7.1f us/step %7.1f Msamp/s  path %s  hot %s" % (sys.argv[1], d["ms_per_step"]*1e3, d["value"]/1e6, d.get("row_sharded_path"), json.dumps(d.get("hot_rows", {}).get("ingress_bytes_per_rank_per_step_at_8_ranks"))))'
for v in "--shard rows --ids zipf --whole-shard off" "--shard rows --ids zipf --whole-shard off --hot-rows 65536" "--shard rows --ids zipf --whole-shard off --dedup off" "--shard rows --ids zipf --whole-shard off --dedup off --hot-rows 65536" "--shard rows --whole-shard off --hot-rows 65536" "--shard rows --ids zipf --nemb 64 --nfeat 100000000 --hot-rows 65536" "--shard rows --ids zipf --nemb 64 --nfeat 100000000"; do
  python bench.py $v --steps 30 --warmup 5 --no-cpu-baseline --no-other-alphas --regime fresh 2>$G/r5_hot_err.txt | python -c "$P" "$v" || tail -5 $G/r5_hot_err.txt
done
} > $G/r5_hot_rows_one_rank.txt 2>&1
cat $G/r5_hot_rows_one_rank.txt
timeout 2400 python -m pytest tests/ -m gpu -x -q > $G/r5_gpu_suite.log 2>&1; echo "suite rc=$?" >> $G/r5_gpu_suite.log
tail -n 8 $G/r5_gpu_suite.log
