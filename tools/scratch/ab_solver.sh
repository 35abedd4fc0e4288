#!/bin/bash
# developer script (GPU box): A/B the current library against arm-net_amd/lib/libarmnet_hip_b.so (a previous build)
# on the headline shape, both weight regimes, several alpha
for alpha in 2.0 1.7 1.5 2.5; do
  for regime in fresh stress; do
    for lib in libarmnet_hip_b.so libarmnet_hip.so; do
      ARMNET_HIP_LIB=$PWD/arm-net_amd/lib/$lib python tools/kbench.py --alpha $alpha --regime $regime --steps 100 2>&1 | tail -1 | sed "s/^/$lib: /"
    done
  done
done
ARMNET_HIP_LIB=$PWD/arm-net_amd/lib/libarmnet_hip_b.so python tools/kbench.py --O 128 --steps 50 --regime stress | tail -1 | sed "s/^/b: /"
python tools/kbench.py --O 128 --steps 50 --regime stress | tail -1 | sed "s/^/new: /"
