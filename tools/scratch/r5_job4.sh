#!/bin/bash
# round 5, GPU call 4: sibling + training tests, training-step timings with the head's GEMMs on the matrix cores vs hipBLASLt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
G=$ROOT/gpurun_out
mkdir -p $G
cd "$ROOT"
timeout 1200 python -m pytest tests/test_siblings.py tests/test_hip_training.py -m gpu -x -q > $G/r5_tests_c.log 2>&1; echo "tests_c rc=$?" >> $G/r5_tests_c.log
tail -n 6 $G/r5_tests_c.log
{
echo "# tools/train_bench.py <B> <alpha>: ARM-Net (39 fields, nemb 16, 32 neurons, 2 x 256 head, Adam), eager and one-hipGraph step"
for gemm in mfma hipblaslt; do
  for B in 65536 4096; do
    ARMNET_HEAD_GEMM=$gemm python tools/train_bench.py $B 1.7 2>&1 | grep -v amdgpu.ids
  done
done
ARMNET_HEAD_GEMM=mfma python tools/train_bench.py 65536 2.0 2>&1 | grep -v amdgpu.ids
echo
echo "# tools/bwd_bench.py: the block's matrix-core backward alone, headline shape"
python tools/bwd_bench.py 2>&1 | grep -v amdgpu.ids
echo
echo "# tools/sibling_train_bench.py (GC-ARM / AFN, 64 neurons): head GEMMs on the matrix cores | hipBLASLt"
for gemm in mfma hipblaslt; do
  echo "## ARMNET_HEAD_GEMM=$gemm"
  ARMNET_HEAD_GEMM=$gemm python tools/sibling_train_bench.py 2>&1 | grep -v amdgpu.ids
done
echo
echo "# tools/train_profile.py: torch.profiler kernel table of 5 training steps at B = 65536"
python tools/train_profile.py 2>&1 | grep -v amdgpu.ids | head -60
} > $G/r5_train_step_times.txt 2>&1
cat $G/r5_train_step_times.txt | head -80
