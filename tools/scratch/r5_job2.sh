#!/bin/bash
# round 5, GPU call 2: the new GPU tests (hot rows, host/device placement), then tools/r5_config_counters.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
G=$ROOT/gpurun_out
mkdir -p $G
cd "$ROOT"
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_host_tensors.py -m gpu -x -q > $G/r5_tests_a.log 2>&1; echo "tests_a rc=$?" >> $G/r5_tests_a.log
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "route or shard or gather" > $G/r5_tests_b.log 2>&1; echo "tests_b rc=$?" >> $G/r5_tests_b.log
tail -4 $G/r5_tests_a.log $G/r5_tests_b.log
bash tools/r5_config_counters.sh > $G/r5_config_counters.log 2>&1
tail -70 $G/r5_config_ablations.txt
