# developer script (GPU box): parity suite, exhaustive scans, then the fused kernels on the block shapes of the
# reference's run.sh (nemb = 10) and BASELINE.json
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/shape_scan.py --bwd 2>&1 | tail -2
python tools/shape_scan_big.py 2>&1 | tail -2
for cfg in "39 16 32 65536" "39 16 128 65536" "39 64 32 65536" "22 32 128 131072" \
           "10 10 256 65536" "3 10 128 65536" "22 10 128 65536" "22 10 64 65536" "39 10 128 65536" "39 10 256 65536" "43 10 32 65536" "43 10 512 65536"; do
  set -- $cfg
  python tools/kbench.py --F $1 --E $2 --O $3 --B $4 2>&1 | tail -1
  python tools/bwd_bench.py --F $1 --E $2 --O $3 --B $4 2>&1 | tail -1
done
