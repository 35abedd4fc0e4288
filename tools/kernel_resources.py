#!/usr/bin/env python3
"""List VGPR / scratch / LDS use of every kernel in libarmnet_hip.so (reads the code-object metadata notes)."""
import re, subprocess, sys, tempfile, os
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "arm-net_amd", "lib", "libarmnet_hip.so")
with tempfile.TemporaryDirectory() as d:
    # the fat binary lives in .hip_fatbin; extract with objcopy then unbundle
    fb = d + "/fb.bin"
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fb])
    data = open(fb, "rb").read()
    # concatenated bundles: split on the magic
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    idx = [m.start() for m in re.finditer(re.escape(magic), data)]
    rows = []
    for n, i in enumerate(idx):
        part = data[i: idx[n + 1] if n + 1 < len(idx) else len(data)]
        pf = f"{d}/b{n}.bin"; open(pf, "wb").write(part)
        co = f"{d}/b{n}.co"
        r = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + pf,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
        if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for blk in txt.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
            name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
            rows.append((name, g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    for r in sorted(rows):
        if filt in r[0]:
            print(f"vgpr={r[1]:>4} sgpr={r[2]:>4} scratch={r[3]:>5} lds={r[4]:>6}  {r[0][:150]}")
