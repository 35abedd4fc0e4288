#!/usr/bin/env python3
"""Instruction census of one fused-kernel instantiation (developer tool; runs in the build container, no GPU).

    python tools/isa_census.py "16,10,2,SOLVE_MICHELOT,0,3,MODEL_ARM" [extra hipcc flags]

Compiles that instantiation of fused_mfma_kernel with -DARMNET_PHASE_TIMING (the s_memtime markers delimit the
phases: 0 staging, 1 MFMA #1, 2 row statistics, 3 solver, 4 weights + MFMA #2, 5 epilogue) and prints, per phase, the
STATIC count of VALU / MFMA / LDS / VMEM / SALU instructions (loop bodies once; `solver-loop` is one evaluation)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1] if len(sys.argv) > 1 else "16,10,2,SOLVE_MICHELOT,0,3,MODEL_ARM"
with tempfile.TemporaryDirectory() as d:
    src = os.path.join(d, "k.hip")
    open(src, "w").write('#include "fused_mfma_kernel.h"\nnamespace armnet {\ntemplate __global__ void '
                         f"fused_mfma_kernel<{args}>(FusedArgs);\n}}\n")
    out = os.path.join(d, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math",
                           "-ffp-contract=off", "-I" + os.path.join(ROOT, "arm-net_amd", "csrc"),
                           "-I" + os.path.join(ROOT, "include"), "-DARMNET_PHASE_TIMING", "-S", "--cuda-device-only",
                           "-o", out, src] + sys.argv[2:], stderr=subprocess.DEVNULL)
    lines = open(out).read().splitlines()


def kind(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return None


phase, seen = "pre", 0
tot = collections.OrderedDict()
ops = collections.defaultdict(collections.Counter)
in_loop3 = False
for ln in lines:
    s = ln.strip()
    if s.startswith(".LBB"):
        in_loop3 = False
    if "Depth=3" in s and "Inner Loop Header" in s:
        in_loop3 = True
    m = re.match(r"([a-z_0-9]+)", s)
    if not m or s.startswith((".", ";")):
        continue
    op = m.group(1)
    if op == "s_memtime":
        seen += 1
        phase = f"after-marker-{seen}"
        continue
    k = kind(op)
    if k is None:
        continue
    key = phase + (" [depth-3 loop]" if in_loop3 else "")
    tot.setdefault(key, collections.Counter())[k] += 1
    if k == "valu":
        ops[key][op] += 1
for key, c in tot.items():
    print(f"{key:34s} valu {c['valu']:4d} mfma {c['mfma']:3d} lds {c['lds']:3d} vmem {c['vmem']:3d} salu {c['salu']:4d} wait {c['wait']:3d}   "
          + " ".join(f"{o}:{n}" for o, n in ops[key].most_common(6)))
m = re.search(r"\.vgpr_count:\s+(\d+)", "\n".join(lines))
sp = re.search(r"\.vgpr_spill_count:\s+(\d+)", "\n".join(lines))
print("vgpr", m.group(1) if m else "?", "spills", sp.group(1) if sp else "?")
