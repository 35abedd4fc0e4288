# developer script (GPU box): A/B two builds of the library on a list of shapes:  run_ab.sh "F E O B" ...
for cfg in "$@"; do
  set -- $cfg
  for lib in libarmnet_hip.so libarmnet_hip_b.so; do
    ARMNET_HIP_LIB=$PWD/arm-net_amd/lib/$lib python tools/kbench.py --F $1 --E $2 --O $3 --B $4 --steps 100 2>&1 | tail -1 | sed "s/^/$lib: /"
  done
done
