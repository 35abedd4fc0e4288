#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; G=gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $G/r6_f16_gpu_suite.log; cat $G/r6_f16_gpu_suite.log
python tools/shape_scan.py 2>&1 | tail -3 > $G/r6_f16_shape_scan.txt; python tools/shape_scan_big.py 2>&1 | tail -3 >> $G/r6_f16_shape_scan.txt; cat $G/r6_f16_shape_scan.txt
python bench.py --nhead 4 --steps 30 --warmup 5 --no-cpu-baseline > $G/r6_f16_bench_k4.json 2> $G/r6_f16_bench_k4.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $G/r6_f16_bench_headline.json 2> $G/r6_f16_bench_headline.err
python - <<'PY'
import json
for f in ("gpurun_out/r6_f16_bench_k4.json", "gpurun_out/r6_f16_bench_headline.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d["value"] / 1e6, 1), round(d["ms_per_step"] * 1e3, 1), round(d["roofline"]["frac"], 3), round(d["full_forward"]["value"] / 1e6, 1),
          {k: round(v["ms_per_step"] * 1e3, 1) for k, v in d["regimes"].items()}, {k: {r: round(x["ms_per_step"] * 1e3, 1) for r, x in v.items() if isinstance(x, dict) and "ms_per_step" in x} for k, v in (d.get("other_alphas") or {}).items()})
PY
