#!/bin/bash
# round 4, GPU call A: routing microbenchmark + the re-plumbed bench.py (median of windows) + its contract tests
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4a
mkdir -p "$OUT"
cd "$ROOT"
./tools/ubench/route_mark > "$OUT/route_mark.txt" 2>&1
./tools/ubench/route_mark 2555904 10000000 >> "$OUT/route_mark.txt" 2>&1
./tools/ubench/route_mark 319488 1000000 >> "$OUT/route_mark.txt" 2>&1
cat "$OUT/route_mark.txt"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench1.json" 2> "$OUT/bench1.err"; echo "bench rc=$?"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench2.json" 2> "$OUT/bench2.err"; echo "bench rc=$?"
python - <<PY
import json
for f in ("bench1","bench2"):
    try:
        d=json.load(open("$OUT/%s.json"%f))
        print(f, "value %.1f M  ms/step %.4f  spread %.3f  windows" % (d["value"]/1e6, d["ms_per_step"], d["value_spread"]), [round(x*1e3,1) for x in d["value_windows_ms"]], d["value_windows_after"], d["clock_settle"], "cold %.1f" % (d["cold_start"]["ms_per_step"]*1e3), "a1.7 %.1f" % (d["other_alphas"]["1.7"]["fresh"]["ms_per_step"]*1e3), "stress %.1f" % (d["regimes"]["stress"]["ms_per_step"]*1e3), "full %.1f M" % (d["full_forward"]["value"]/1e6), d["cpu_baseline"]["value"], d["cpu_baseline"]["aten_chain"])
    except Exception as e:
        print(f, "FAILED", e); print(open("$OUT/%s.err"%f).read()[-3000:])
PY
timeout 1500 python -m pytest tests/test_bench_contract.py -x -q -m gpu 2>&1 | tail -15
