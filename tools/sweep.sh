#!/bin/bash
# The measurement matrix of SURVEY.md §8(d) on one GPU: alpha x weight regime x id distribution + configs 3..5.
P='import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d["roofline"]; c=d["config"]["workload"]
print("%-118s | %7.1f Msamp/s | %7.1f us | hbm-frac %.3f | %5.1f TF | full fwd %7.1f Msamp/s" % (c[:118], d["value"]/1e6, r["kernel_ms"]*1e3, r["frac"], r["folded_tflops"], d["full_forward"]["value"]/1e6))'
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-alphas "$@" 2>/dev/null | python -c "$P"; }
for al in 2.0 1.7 1.5 1.0; do for r in fresh stress; do run --alpha $al --regime $r; done; done
run --alpha 2.5
run --ids zipf
run --ids zipf --regime stress
run --nhead 4
run --nhead 4 --regime stress --alpha 1.7
run --nemb 64 --nfeat 10000000
run --nemb 64 --nfeat 100000000 --shard rows
run --nhead 4 --nemb 32 --nfield 22 --nfeat 2000000 --batch 131072
run --nhead 4 --nemb 32 --nfield 22 --nfeat 2000000 --batch 131072 --ensemble
run --shard rows
run --batch 8192
run --batch 262144
