#!/bin/bash
# round 4, GPU call B: parity of the new pieces, routing before / after, nemb > 64 on the matrix cores against the generic kernel
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4b
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "route or shape_sweep or g13 or sharded or nfield" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_hip_training.py -x -q -m gpu -k "mfma_backward or h4" 2>&1 | tail -8
timeout 600 python tools/route_bench.py 2>&1 | tee "$OUT/route_bench.txt"
{
for cfg in "10 100 10 1.7" "10 100 10 2.0" "39 96 32 2.0" "22 72 32 2.0" "39 128 32 2.0" "10 128 128 1.7"; do
  set -- $cfg
  python tools/kbench.py --F $1 --E $2 --O $3 --alpha $4 --nfeat 1000000 --flags 0 0x4
done
} 2>&1 | tee "$OUT/kbench_e128.txt"
python tools/bwd_bench.py --help > /dev/null 2>&1
timeout 900 python bench.py --shard rows --whole-shard off --no-cpu-baseline --no-other-alphas --regime fresh --steps 20 --warmup 5 > "$OUT/bench_rows.json" 2> "$OUT/bench_rows.err"; echo "rows rc=$?"
timeout 900 python bench.py --shard rows --whole-shard off --dedup off --no-cpu-baseline --no-other-alphas --regime fresh --steps 20 --warmup 5 > "$OUT/bench_rows_nodedup.json" 2> "$OUT/bench_rows_nodedup.err"; echo "rows rc=$?"
timeout 900 python bench.py --shard rows --nemb 64 --nfeat 100000000 --no-cpu-baseline --no-other-alphas --regime fresh --steps 20 --warmup 5 > "$OUT/bench_rows_c4.json" 2> "$OUT/bench_rows_c4.err"; echo "rows c4 rc=$?"
python - <<PY
import json
for f in ("bench_rows","bench_rows_nodedup","bench_rows_c4"):
    try:
        d=json.load(open("$OUT/%s.json"%f))
        print(f, "ms/step %.4f  value %.1f M  windows" % (d["ms_per_step"], d["value"]/1e6), [round(x*1e3,1) for x in d["value_windows_ms"]], d.get("row_sharded_overflow"))
    except Exception as e:
        print(f, "FAILED", e); print(open("$OUT/%s.err"%f).read()[-2000:])
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_rows" -- python $ROOT/bench.py --shard rows --whole-shard off --no-cpu-baseline --no-other-alphas --regime fresh --steps 20 --warmup 5 --settle-ms 0 --in-flight 1 > "$OUT/prof_rows.log" 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$OUT/prof_rows/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
    for r in rows[:14]:
        print("%-90s calls %6s avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3))
PY
rm -rf "$OUT/prof_rows"/*/*.db
