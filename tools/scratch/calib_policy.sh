#!/bin/bash
# round 3: does a cache-policy modifier on the row loads change what a 64-byte row costs at the fabric?  (GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/calib_pol; mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o "$ROOT/tools/ubench/gather_calib" "$ROOT/tools/ubench/gather_calib.hip" || exit 1
cd /tmp && export TMPDIR=/tmp
ROWS=2555904
: > "$OUT/summary.txt"
for nf in 1000000 64000000; do
for pol in 0 1 2 3 4 5 6 7; do
  echo "=== policy $pol nfeat $nf" >> "$OUT/summary.txt"
  "$ROOT/tools/ubench/gather_calib" 64 $nf $ROWS 20 $pol >> "$OUT/summary.txt" 2>&1
  for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/p${pol}_$nf/$tag" -- "$ROOT/tools/ubench/gather_calib" 64 $nf $ROWS 5 $pol > "$OUT/p${pol}_$nf.$tag.log" 2>&1
    python3 - "$OUT/p${pol}_$nf/$tag" >> "$OUT/summary.txt" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "only" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
for c in sorted(acc):
    print(f"    {c:32s} per row {acc[c] / max(1, len(n[c])) / 2555904:10.4f}")
PY
  done
done
done
cat "$OUT/summary.txt"
