#!/bin/bash
# Collect the rocprofv3 evidence for the fused kernel on the GPU box (run through gpurun):
#   pass 0: kernel trace + stats (timings)          pass 1..n: PMC counters, one group per process
# usage: tools/profile.sh <outdir-under-gpurun_out> -- <command...>
set -u
OUT=$1; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/gpurun_out/$OUT"
cd /tmp && export TMPDIR=/tmp
# PMC_EXTRA: extra arguments for the command in the counter passes only (counters do not depend on the clocks, so
# bench.py's untimed clock-settling pre-run — thousands of dispatches — is switched off there: --settle-ms 0)
run() { # name, extra args...
  local name=$1; shift
  local extra=()
  if [ "$name" != trace ] && [ -n "${PMC_EXTRA:-}" ]; then read -r -a extra <<< "$PMC_EXTRA"; fi
  rocprofv3 "$@" --kernel-trace --output-format csv -d "$ROOT/gpurun_out/$OUT/$name" -- "${CMD[@]}" "${extra[@]}" > "$ROOT/gpurun_out/$OUT/$name.log" 2>&1
}
CMD=("$@")
run trace --stats
if [ -n "${PROFILE_LIGHT:-}" ]; then   # PROFILE_LIGHT=1: timings + the first SQ pass only (second weight regime)
  run pmc_sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
  exit 0
fi
run pmc_sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run pmc_sq2 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_WAIT_INST_LDS
run pmc_sq3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL
run pmc_fetch --pmc FETCH_SIZE
run pmc_rdreq --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run pmc_wrreq --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run pmc_write --pmc WRITE_SIZE
run pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run pmc_grbm --pmc GRBM_GUI_ACTIVE GRBM_COUNT
find "$ROOT/gpurun_out/$OUT" -name "*.csv" | head -40
