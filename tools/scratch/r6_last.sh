#!/bin/bash
# round 6, last call: the full GPU suite, smoke, the default bench line, the RCCL path at world 1 — on the final tree
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; G=gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $G/r6_last_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $G/r6_last_smoke.log 2>&1
python bench.py > $G/r6_last_bench.json 2> $G/r6_last_bench.err
ARMNET_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $G/r6_last_bench_forced_dist.json 2> $G/r6_last_bench_forced_dist.err
cat $G/r6_last_gpu_suite.log; tail -2 $G/r6_last_smoke.log; python - <<'PY'
import json
for f in ("gpurun_out/r6_last_bench.json", "gpurun_out/r6_last_bench_forced_dist.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d.get("value"), d.get("ms_per_step"), (d.get("roofline") or {}).get("frac"), (d.get("full_forward") or {}).get("value"), d.get("error"))
    except Exception as e:
        print(f, "unreadable", e)
PY
