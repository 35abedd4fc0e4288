#!/bin/bash
# quick PMC pass (SQ counters only) for a command; usage: tools/profile_quick.sh <outdir> -- <cmd...>
set -u
OUT=$1; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/gpurun_out/$OUT"
cd /tmp && export TMPDIR=/tmp
CMD=("$@")
run() { local name=$1; shift; rocprofv3 "$@" --kernel-trace --output-format csv -d "$ROOT/gpurun_out/$OUT/$name" -- "${CMD[@]}" > "$ROOT/gpurun_out/$OUT/$name.log" 2>&1; }
run trace --stats
run pmc_sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run pmc_sq2 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_WAIT_INST_LDS
run pmc_grbm --pmc GRBM_GUI_ACTIVE
