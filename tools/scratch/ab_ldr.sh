#!/bin/bash
# developer script (GPU box): the loader / consumer build (make EXTRA=-DARMNET_LDR OUT=../lib/libarmnet_hip_b.so) against
# the product library: parity tests of the headline family first, then timings
B=$PWD/arm-net_amd/lib/libarmnet_hip_b.so
ARMNET_HIP_LIB=$B timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_headline.py -m gpu -x -q 2>&1 | tail -5
for alpha in 2.0 1.7; do
  for regime in fresh stress; do
    for lib in libarmnet_hip.so libarmnet_hip_b.so; do
      ARMNET_HIP_LIB=$PWD/arm-net_amd/lib/$lib timeout 300 python tools/kbench.py --alpha $alpha --regime $regime --steps 100 2>&1 | tail -1 | sed "s/^/$lib a=$alpha $regime: /"
    done
  done
done
for lib in libarmnet_hip.so libarmnet_hip_b.so; do
  ARMNET_HIP_LIB=$PWD/arm-net_amd/lib/$lib timeout 300 python tools/kbench.py --O 128 --steps 50 2>&1 | tail -1 | sed "s/^/$lib O=128: /"
  ARMNET_HIP_LIB=$PWD/arm-net_amd/lib/$lib timeout 300 python bench.py --steps 50 --warmup 10 --no-other-alphas --cpu-seconds 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value']/1e6, d['regimes'], d['full_forward'])" | sed "s/^/$lib: /"
done
