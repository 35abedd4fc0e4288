# developer script (GPU box): parity suite, then the fused kernel on the block shapes of the reference's run.sh
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
if [ -f arm-net_amd/lib/libarmnet_hip_old.so ]; then
  for i in 1 2 3; do
    ARMNET_HIP_LIB=$PWD/arm-net_amd/lib/libarmnet_hip_old.so python tools/kbench.py --steps 100 2>&1 | tail -1 | sed 's/^/old: /'
    python tools/kbench.py --steps 100 2>&1 | tail -1 | sed 's/^/new: /'
  done
  for cfg in "39 64 32 65536" "22 32 32 65536" "39 16 128 65536"; do
    set -- $cfg
    ARMNET_HIP_LIB=$PWD/arm-net_amd/lib/libarmnet_hip_old.so python tools/kbench.py --F $1 --E $2 --O $3 --B $4 2>&1 | tail -1 | sed 's/^/old: /'
    python tools/kbench.py --F $1 --E $2 --O $3 --B $4 2>&1 | tail -1 | sed 's/^/new: /'
  done
fi
