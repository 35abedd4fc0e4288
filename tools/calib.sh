#!/bin/bash
# round 3: calibrate FETCH_SIZE for random row gathers (run through gpurun): builds tools/ubench/gather_calib and runs it
# under rocprofv3 --pmc for 64- / 128- / 256-byte rows, on a MALL-resident (64 MB) and an HBM-resident (4 GB) table, plus
# the coalesced-stream control.  Output: gpurun_out/calib/summary.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/calib
mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o "$ROOT/tools/ubench/gather_calib" "$ROOT/tools/ubench/gather_calib.hip" || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
ROWS=2555904
: > "$OUT/summary.txt"
for cfg in "64 1000000" "128 1000000" "128 500000" "64 64000000" "128 32000000" "256 16000000" "stream 64000000"; do
  set -- $cfg
  name="rb$1_nf$2"
  echo "=== $cfg" >> "$OUT/summary.txt"
  "$ROOT/tools/ubench/gather_calib" $1 $2 $ROWS 20 >> "$OUT/summary.txt" 2>&1
  for grp in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$name/$tag" -- "$ROOT/tools/ubench/gather_calib" $1 $2 $ROWS 5 > "$OUT/$name.$tag.log" 2>&1
    python3 - "$OUT/$name/$tag" >> "$OUT/summary.txt" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "only" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
for c in sorted(acc):
    print(f"    {c:32s} per dispatch {acc[c] / max(1, len(n[c])):16.1f}   per row {acc[c] / max(1, len(n[c])) / 2555904:10.4f}")
if not acc:
    print("    (no counters collected: " + sys.argv[1].split('/')[-1] + ")")
PY
  done
done
cat "$OUT/summary.txt"
