#!/bin/bash
# round 5: A/B of the closed-form shortcut for all-in-support rows (alpha = 2): product | -DARMNET_NO_DENSE
cd ${GRAFT_REPO_ROOT:-$(pwd)}
P='import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print("%-34s fresh %7.2f us  trained-like %7.2f us  full fwd %6.1f M  windows %s" % (sys.argv[1], d["ms_per_step"]*1e3, d["regimes"]["stress"]["ms_per_step"]*1e3, d["full_forward"]["value"]/1e6, [round(w*1e3,1) for w in d["value_windows_ms"]]))'
{
for rep in 1 2; do
for lib in lib/libarmnet_hip.so lib/exp/libarmnet_nodense.so; do
  ARMNET_HIP_LIB=$PWD/arm-net_amd/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-alphas 2>/dev/null | python -c "$P" "$lib"
done; done
for lib in lib/libarmnet_hip.so lib/exp/libarmnet_nodense.so; do
  ARMNET_HIP_LIB=$PWD/arm-net_amd/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-alphas --ids zipf 2>/dev/null | python -c "$P" "$lib zipf"
  ARMNET_HIP_LIB=$PWD/arm-net_amd/$lib python tools/kbench.py --alpha 2.0 --regime fresh --steps 100 2>&1 | tail -1 | sed "s|^|$lib kbench: |"
  ARMNET_HIP_LIB=$PWD/arm-net_amd/$lib python tools/kbench.py --alpha 2.0 --regime stress --steps 100 2>&1 | tail -1 | sed "s|^|$lib kbench: |"
  ARMNET_HIP_LIB=$PWD/arm-net_amd/$lib python tools/kbench.py --alpha 2.0 --regime fresh --F 10 --E 10 --O 32 --steps 100 2>&1 | tail -1 | sed "s|^|$lib kbench: |"
done
} > gpurun_out/r5_dense_ab.txt 2>&1
cat gpurun_out/r5_dense_ab.txt
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_headline.py -m gpu -x -q 2>&1 | tail -2
