#!/bin/bash
# round 4, GPU call C: full GPU suite on the new library + routing profile + row-sharded bench lines
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4c
mkdir -p "$OUT"
cd "$ROOT"
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee "$OUT/gpu_tests_tail.txt"
timeout 600 python tools/route_bench.py 2>&1 | tee "$OUT/route_bench.txt"
for v in "rows_dedup --whole-shard off" "rows_nodedup --whole-shard off --dedup off" "rows_whole"; do
  set -- $v; name=$1; shift
  timeout 900 python bench.py --shard rows "$@" --no-cpu-baseline --no-other-alphas --regime fresh --steps 20 --warmup 5 > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; echo "$name rc=$?"
done
python - <<PY
import json
for f in ("rows_dedup","rows_nodedup","rows_whole"):
    try:
        d=json.load(open("$OUT/bench_%s.json"%f))
        print(f, "ms/step %.4f  value %.1f M  full %.1f M windows" % (d["ms_per_step"], d["value"]/1e6, d["full_forward"]["value"]/1e6), [round(x*1e3,1) for x in d["value_windows_ms"]], d.get("row_sharded_overflow"))
    except Exception as e:
        print(f, "FAILED", e); print(open("$OUT/bench_%s.err"%f).read()[-2000:])
PY
cd /tmp && export TMPDIR=/tmp
for v in "dedup --whole-shard off" "nodedup --whole-shard off --dedup off"; do
  set -- $v; name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$name" -- python $ROOT/bench.py --shard rows "$@" --no-cpu-baseline --no-other-alphas --regime fresh --steps 20 --warmup 5 --settle-ms 0 --in-flight 1 > "$OUT/prof_$name.log" 2>&1
  echo "== kernel stats, bench.py --shard rows $*" | tee -a "$OUT/prof_summary.txt"
  python - <<PY | tee -a "$OUT/prof_summary.txt"
import csv, glob
for f in glob.glob("$OUT/prof_$name/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
    for r in rows[:14]:
        print("%-100s calls %6s avg %9.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3))
PY
  rm -rf "$OUT/prof_$name"/*/*.db "$OUT/prof_$name"/*/*kernel_trace.csv
done
