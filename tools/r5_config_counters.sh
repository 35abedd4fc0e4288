#!/bin/bash
# round 5 (GPU box, through gpurun; verdict item 2): what bounds the fused block at BASELINE.json configs[2] / [3] / [4]
# (K = 4 -> 128 neurons; nemb 64; Avazu shape, nemb 32, 128 neurons, B = 131 072)?  Per shape: rocprofv3 kernel trace +
# one --pmc pass per counter group over tools/kbench.py (the block alone, HIP events), then the in-kernel ablations of a
# -DARMNET_DEV_FLAGS build (0x400 no stores, 0x200 cache-resident rows, 0x800 no MFMA, 0x100 no solver iterations) and
# the s_memtime phase shares of a -DARMNET_PHASE_TIMING build.
# needs: arm-net_amd/lib/exp/libarmnet_dev.so, libarmnet_phase.so (make ... EXTRA=-DARMNET_DEV_FLAGS / -DARMNET_PHASE_TIMING)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
G=$ROOT/gpurun_out
mkdir -p "$G"
cd /tmp && export TMPDIR=/tmp
prof() { # outdir, kbench args...
  local out=$1; shift
  mkdir -p "$G/$out"
  local cmd=(python "$ROOT/tools/kbench.py" "$@" --steps 40)
  # every pass under its own timeout: one counter group that hangs (round 5: the TCP_* group did, for 24 minutes) must not
  # take the rest of the call with it
  pass() { local name=$1; shift; timeout 180 rocprofv3 "$@" --kernel-trace --output-format csv -d "$G/$out/$name" -- "${cmd[@]}" > "$G/$out/$name.log" 2>&1 || echo "pass $name of $out: rc $?" >> "$G/r5_config_counters_failures.txt"; }
  pass trace --stats
  pass pmc_sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
  pass pmc_sq2 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_WAIT_INST_LDS
  pass pmc_sq3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL
  pass pmc_rdreq --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
  pass pmc_wrreq --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum
  pass pmc_write --pmc WRITE_SIZE
  pass pmc_fetch --pmc FETCH_SIZE
  pass pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
  pass pmc_grbm --pmc GRBM_GUI_ACTIVE GRBM_COUNT
  { echo "# tools/r5_config_counters.sh: python tools/kbench.py $* --steps 40  (1 x MI355X; per-dispatch means)"
    python "$ROOT/tools/prof_summary.py" "$G/$out" fused; } > "$G/${out}_rocprof_summary.txt" 2>&1
}
cd "$ROOT"
{
echo "# in-kernel ablations (libarmnet_dev.so): flags 0 | 0x400 no stores | 0x200 cache-resident rows | 0x600 both | 0x800 no MFMA | 0x100 no solver iterations | 0xf00 all"
for cfg in "39 16 128 65536 1000000" "39 64 32 65536 10000000" "22 32 128 131072 2000000" "39 16 32 65536 1000000"; do
  set -- $cfg
  for regime in fresh stress; do
    ARMNET_HIP_LIB=$ROOT/arm-net_amd/lib/exp/libarmnet_dev.so python tools/kbench.py --F $1 --E $2 --O $3 --B $4 --nfeat $5 --regime $regime --steps 60 --flags 0 0x400 0x200 0x600 0x800 0xe00 0x100 0xf00 2>&1 | grep flags=
  done
done
echo
echo "# phase shares (libarmnet_phase.so, s_memtime sums over all waves), headline shape with 128 neurons"
for regime in fresh stress; do
  ARMNET_HIP_LIB=$ROOT/arm-net_amd/lib/exp/libarmnet_phase.so python tools/phase_timing.py 2.0 $regime 0 128 2>&1 | grep -v amdgpu.ids
done
ARMNET_HIP_LIB=$ROOT/arm-net_amd/lib/exp/libarmnet_phase.so python tools/phase_timing.py 2.0 fresh 0 32 2>&1 | grep -v amdgpu.ids
} > $G/r5_config_ablations.txt 2>&1
cd /tmp
if [ -z "${SKIP_CONFIGS:-}" ]; then
prof r5_config4 --F 39 --E 64 --O 32 --B 65536 --nfeat 10000000
prof r5_config5 --F 22 --E 32 --O 128 --B 131072 --nfeat 2000000
prof r5_config3 --F 39 --E 16 --O 128 --B 65536
fi
cd "$ROOT"
cat $G/r5_config3_rocprof_summary.txt; tail -60 $G/r5_config_ablations.txt
