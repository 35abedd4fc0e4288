"""Measurements bench.py QUOTES rather than takes: counter traffic committed under profiles/ for this kernel source, the access
pattern's ceiling (committed and live)."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def kernel_src_sha():
    """hash of the fused forward kernel's sources: PMC traffic committed under profiles/ is only quoted for the
    kernel it was measured on"""
    h = hashlib.sha256()
    for f in ("fused_mfma_kernel.h", "fused_mfma.hip", "armnet_common.h"):
        with open(os.path.join(ROOT, "arm-net_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def committed_measurements(a, regime):
    """HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes) and the access pattern's own ceiling
    (tools/ubench/gather_stream) as committed under profiles/ for exactly this workload AND this kernel source;
    counters cannot be read from inside the timed process.  (traffic, traffic_tag, pattern_ceiling_us, its source)"""
    key = (f"nfield={a.nfield} nemb={a.nemb} nhid={a.nhid} nhead={a.nhead} B={a.batch} alpha={a.alpha} "
           f"ids={a.ids} regime={regime} rotate={a.rotate}")
    traffic = tag = ceil_us = ceil_src = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            for t in json.load(f)["entries"]:
                if t["workload"] == key and t["kernel_src_sha"] == kernel_src_sha():
                    traffic, tag = t["traffic_bytes_per_launch"], f"{t['source']} @ kernel_src_sha {t['kernel_src_sha']}"
    except Exception:
        pass
    try:
        with open(os.path.join(ROOT, "profiles", "access_pattern_ceiling.json")) as f:
            for t in json.load(f)["entries"]:
                if t["workload"] == f"nfield={a.nfield} nemb={a.nemb} nhid={a.nhid} nhead={a.nhead} B={a.batch}":
                    ceil_us, ceil_src = t["us"], t["source"]
    except Exception:
        pass
    return traffic, tag, ceil_us, ceil_src


def live_pattern_ceiling(a):
    """tools/ubench/gather_stream (built by __graft_entry__.build()) run NOW on this device: the fused block's memory
    traffic and nothing else, rotating over 4 batches like the timed steps.  Only for the workload it implements (the
    headline shape); None when the binary is not there."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tools", "ubench", "gather_stream")
    if not (os.path.exists(exe) and (a.nfield, a.nemb, a.nhid, a.nhead, a.batch, a.nfeat) == (39, 16, 32, 1, 65536, 1_000_000)):
        return None
    try:
        out = subprocess.run([exe, str(max(1, a.rotate)), "quick"], capture_output=True, timeout=60).stdout.decode()
        us = [float(m) for m in re.findall(r":\s*([0-9.]+) us", out)]
        return min(us) if us else None
    except Exception:
        return None
