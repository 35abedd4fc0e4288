L=$PWD/arm-net_amd/lib
for regime in fresh stress; do
ARMNET_HIP_LIB=$L/exp/libarmnet_dev.so python tools/kbench.py --regime $regime --steps 100 --flags 0 0x600 0xE00 0x700 0xF00 2>&1 | grep "us"
done
ARMNET_HIP_LIB=$L/exp/libarmnet_dev.so python tools/kbench.py --O 128 --steps 50 --flags 0 0x600 0xE00 0x700 0xF00 2>&1 | grep "us"
