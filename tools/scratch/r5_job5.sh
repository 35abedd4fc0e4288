#!/bin/bash
# round 5, GPU call 5: sibling / training tests again, the exhaustive forward shape scans on the first-order-finish kernels,
# one default bench.py line (sanity of the new settle rule and of the line's shape)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
G=$ROOT/gpurun_out
mkdir -p $G
cd "$ROOT"
timeout 1200 python -m pytest tests/test_siblings.py tests/test_hip_training.py -m gpu -x -q > $G/r5_tests_c.log 2>&1; echo "tests_c rc=$?" >> $G/r5_tests_c.log
tail -n 4 $G/r5_tests_c.log
{ timeout 1500 python tools/shape_scan.py 2>&1 | grep -v amdgpu.ids | tail -20
  timeout 1500 python tools/shape_scan.py --wide 2>&1 | grep -v amdgpu.ids | tail -20
  timeout 900 python tools/shape_scan_big.py 2>&1 | grep -v amdgpu.ids | tail -8; } > $G/r5_shape_scans.txt 2>&1
cat $G/r5_shape_scans.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $G/r5_bench_n1_first.json 2> $G/r5_bench_n1_first.err; tail -c 1500 $G/r5_bench_n1_first.json; tail -3 $G/r5_bench_n1_first.err
