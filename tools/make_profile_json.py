#!/usr/bin/env python3
"""Turn a tools/profile.sh output directory (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py) and a
tools/ubench/gather_stream log into the two JSON files bench.py quotes: profiles/pmc_traffic.json (HBM bytes per
launch of the fused kernel, tagged with the hash of the kernel sources it was measured on) and
profiles/access_pattern_ceiling.json.

    python tools/make_profile_json.py <profile dir> <gather_stream log> <round tag>"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def counter_mean(d, name, kernel_sub):
    vals = {}
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:                      # the newest pass only (gpurun merges earlier runs into the same directory)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and kernel_sub in r["Kernel_Name"]:
                vals.setdefault(r["Dispatch_Id"], 0.0)
                vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
    return sum(vals.values()) / max(1, len(vals)), len(vals)


def main(prof_dir, gs_log, tag):
    import bench
    fetch_kib, n1 = counter_mean(os.path.join(prof_dir, "pmc_fetch"), "FETCH_SIZE", "fused_mfma_kernel")
    write_kib, n2 = counter_mean(os.path.join(prof_dir, "pmc_write"), "WRITE_SIZE", "fused_mfma_kernel")
    rdreq, n3 = counter_mean(os.path.join(prof_dir, "pmc_rdreq"), "TCC_EA0_RDREQ_sum", "fused_mfma_kernel")
    rd32, _ = counter_mean(os.path.join(prof_dir, "pmc_rdreq"), "TCC_EA0_RDREQ_32B_sum", "fused_mfma_kernel")
    B, F, E, O = 65536, 39, 16, 32
    # Calibration (profiles/r03_fetch_size_calibration_gather_rows.txt, tools/ubench/gather_calib.hip): on gfx950 one
    # fabric read request of the "large" class moves a 128-byte line — a coalesced stream issues 0.5 requests per 64
    # bytes, a random 64-BYTE row ONE request, a random 128-byte row ONE request, a 256-byte row two — and FETCH_SIZE
    # tallies every such request at 64 bytes (TCC_EA0_RDREQ_32B, the small class, stays 0).  The bytes the fabric moves
    # are therefore 128 x TCC_EA0_RDREQ (= 2 x FETCH_SIZE) for EVERY read of this kernel, the half-used lines of the
    # 64-byte row gathers included.  WRITE_SIZE matched the written bytes to 0.02 % in rounds 1-2 (1-KiB contiguous
    # stores per wave instruction) and is taken as is.
    read_bytes = 128.0 * rdreq if rdreq else 2.0 * fetch_kib * 1024
    entry = {
        "workload": f"nfield={F} nemb={E} nhid={O} nhead=1 B={B} alpha=2.0 ids=uniform regime=fresh rotate=4",
        "kernel_src_sha": bench.kernel_src_sha(),
        "source": f"profiles/{tag}_bench_n1_rocprof_summary.txt (rocprofv3 --pmc FETCH_SIZE / TCC_EA0_RDREQ_sum / WRITE_SIZE, "
                  f"separate passes, {n1}/{n3}/{n2} dispatches of `python bench.py --steps 20 --warmup 5 --no-cpu-baseline "
                  f"--regime fresh --in-flight 1 --settle-ms 0`)",
        "fetch_size_kib": fetch_kib, "write_size_kib": write_kib, "tcc_ea0_rdreq": rdreq, "tcc_ea0_rdreq_32b": rd32,
        "fetch_bytes_raw": fetch_kib * 1024, "write_bytes": write_kib * 1024,
        "read_bytes_at_128_per_request": read_bytes,
        "traffic_bytes_per_launch": int(read_bytes + write_kib * 1024),
        "algorithmic_bytes_per_launch": B * (F * (12 + 4 * E) + 4 * O * E),
        "_comment": "read traffic = 128 bytes x TCC_EA0_RDREQ (calibrated: profiles/r03_fetch_size_calibration_gather_rows.txt): "
                    "a random 64-byte embedding row costs one 128-byte request, exactly like a 128-byte row, so the "
                    "2.56 M row gathers of a launch move 327 MB of lines for 164 MB of rows; the steps rotate over 4 "
                    "distinct batches (660 MB), so only the 64 MB table can be served from the Infinity Cache (the "
                    "counters are fabric-side: cache hits included)",
    }
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump({"entries": [entry]}, f, indent=1)
    us = None
    for line in open(gs_log):
        m = re.search(r"blocks=\s*4096 store=1\s*:\s*([0-9.]+) us", line)
        if m:
            us = float(m.group(1)) if us is None else min(us, float(m.group(1)))
    ceil = {"entries": [{"workload": f"nfield={F} nemb={E} nhid={O} nhead=1 B={B}", "us": us,
                         "source": f"profiles/{tag}_ubench_gather_stream_memory_floor.txt (tools/ubench/gather_stream: the "
                                   f"fused kernel's memory traffic and nothing else — 39 ids + 39 values + 39 random "
                                   f"64-byte rows read, 2 KiB written per sample)"}]}
    with open(os.path.join(ROOT, "profiles", "access_pattern_ceiling.json"), "w") as f:
        json.dump(ceil, f, indent=1)
    print(json.dumps(entry, indent=1))
    print(ceil)


if __name__ == "__main__":
    main(*sys.argv[1:4])
