#!/bin/bash
# round 4: is the full-size config 4 (nfeat 100 M x nemb 64 = 25.6 GB) bound by address translation?  (run through gpurun)
#   1. tools/ubench/gather_tlb: random 256-byte / 64-byte row gathers over tables of 0.25 ... 64 GB, uniform / per-XCD slab /
#      sorted / windowed ids, hipMalloc vs contiguous allocation: time per 2.56 M rows
#   2. the same under rocprofv3 --pmc with the UTCL1 / UTCL2 counters for a small and the full-size table
#   3. bench.py --nemb 64 at nfeat 10 M and 100 M under the same counters
# Output: gpurun_out/tlb/summary.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/tlb
mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o "$ROOT/tools/ubench/gather_tlb" "$ROOT/tools/ubench/gather_tlb.hip" || exit 1
cd /tmp && export TMPDIR=/tmp
S=$OUT/summary.txt
: > "$S"
G=$ROOT/tools/ubench/gather_tlb
echo "### sweep: table size, 256-byte rows, uniform ids" >> "$S"
for gb in 0.25 1 2.56 4 8 12 16 25.6 48 96; do timeout 120 $G 256 $gb 0 >> "$S" 2>&1; done
echo "### 64-byte rows, uniform ids" >> "$S"
for gb in 0.064 1 6.4 25.6; do timeout 120 $G 64 $gb 0 >> "$S" 2>&1; done
echo "### 25.6 GB, 256-byte rows: xcd-slab / sorted / windows / contiguous allocation" >> "$S"
timeout 120 $G 256 25.6 1 >> "$S" 2>&1
timeout 120 $G 256 25.6 2 >> "$S" 2>&1
for w in 64 512 2048 4096 8192; do timeout 120 $G 256 25.6 3 0 2555904 10 $w >> "$S" 2>&1; done
timeout 120 $G 256 25.6 0 1 >> "$S" 2>&1
timeout 120 $G 256 96 1 >> "$S" 2>&1
pmc() {  # name, then the command
  name=$1; shift
  for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" "TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
    tag=$(echo $grp | tr ' ' '+' | cut -c1-60)
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmc/$name/$tag" -- "$@" > "$OUT/pmc_$name.$tag.log" 2>&1
    python3 - "$OUT/pmc/$name/$tag" "$name" >> "$S" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not ("gather_rows" in k or "fused_mfma_kernel" in k):
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
for c in sorted(acc):
    print(f"  [{sys.argv[2]}] {c:48s} per dispatch {acc[c] / max(1, len(n[c])):16.1f}  ({len(n[c])} dispatches)")
if not acc:
    print(f"  [{sys.argv[2]}] (no counters collected: " + sys.argv[1].split('/')[-1] + ")")
PY
  done
}
echo "### counters: gather_tlb 256-byte rows, 2.56 GB vs 25.6 GB uniform, 25.6 GB xcd-slab" >> "$S"
pmc g2p56 $G 256 2.56 0 0 2555904 3
pmc g25p6 $G 256 25.6 0 0 2555904 3
pmc g25p6slab $G 256 25.6 1 0 2555904 3
cd "$ROOT"
echo "### bench.py fused block, nemb 64: nfeat 10 M vs 100 M" >> "$S"
for nf in 10000000 100000000; do
  timeout 600 python bench.py --nemb 64 --nfeat $nf --regime fresh --no-cpu-baseline --no-other-alphas --steps 20 --warmup 5 > "$OUT/bench_nf$nf.json" 2> "$OUT/bench_nf$nf.err"
  python3 -c "
import json,sys
d=json.load(open('$OUT/bench_nf$nf.json'))
print('  bench nemb=64 nfeat=$nf: %.1f us/step, frac %.3f, full_forward %.1f M/s' % (d['ms_per_step']*1e3, d['roofline']['frac'], d['full_forward']['value']/1e6))
" >> "$S" 2>&1
done
cd /tmp
pmc b10m python $ROOT/bench.py --nemb 64 --nfeat 10000000 --regime fresh --no-cpu-baseline --no-other-alphas --steps 10 --warmup 2 --settle-ms 0 --in-flight 1
pmc b100m python $ROOT/bench.py --nemb 64 --nfeat 100000000 --regime fresh --no-cpu-baseline --no-other-alphas --steps 10 --warmup 2 --settle-ms 0 --in-flight 1
cd "$ROOT"
echo "### headline bench.py, default flags (this box)" >> "$S"
timeout 600 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python3 -c "
import json
d=json.load(open('$OUT/bench_default.json'))
print('  headline: %.1f us/step value %.1f M/s frac %.3f cold %.1f us; a1.7 fresh %.1f us; stress %.1f us' % (d['ms_per_step']*1e3, d['value']/1e6, d['roofline']['frac'], d['cold_start']['ms_per_step']*1e3, d['other_alphas']['1.7']['fresh']['ms_per_step']*1e3, d['regimes']['stress']['ms_per_step']*1e3))
" >> "$S" 2>&1
rm -rf "$OUT/pmc"/*/*/*/*.db 2>/dev/null
cat "$S"
