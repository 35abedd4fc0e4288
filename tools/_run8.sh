L=$PWD/arm-net_amd/lib
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/shape_scan.py 2>&1 | tail -1
for rep in 1 2; do for lib in exp/libarmnet_nco.so libarmnet_hip.so; do
  ARMNET_HIP_LIB=$L/$lib python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['regimes']; oa=d['other_alphas']; print('fresh %.1f us stress %.1f us full %.1f M inflight %.1f | a1.7 %.1f/%.1f' % (r['fresh']['kernel_ms']*1e3, r['stress']['kernel_ms']*1e3, d['full_forward']['value']/1e6, d['batches_in_flight']['value']/1e6, oa['1.7']['fresh']['ms_per_step']*1e3, oa['1.7']['stress']['ms_per_step']*1e3))" | sed "s|^|$lib: |"
done; done
for F in 3 10 22 43; do for ar in "2.0 fresh" "2.0 stress"; do set -- $ar
for lib in exp/libarmnet_nco.so libarmnet_hip.so; do
  ARMNET_HIP_LIB=$L/$lib python tools/kbench.py --F $F --E 10 --O 32 --alpha $1 --regime $2 --steps 100 2>&1 | tail -1 | awk -v l=$lib '{print l, $2, $3, $5, $6, $7, $8, $9}'
done; done; done
