#!/bin/bash
# round 6: shape scans on the final kernel sources (forward / wide / backward / big) + the siblings' scans incl. the nemb 65..128 family
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; G=gpurun_out/r6_shape_scans_final.txt; : > $G
for cmd in "tools/shape_scan.py" "tools/shape_scan.py --wide" "tools/shape_scan.py --bwd" "tools/shape_scan_big.py" \
           "tools/sibling_fwd_scan.py" "tools/sibling_fwd_scan.py --wide" "tools/sibling_bwd_scan.py --wide" "tools/sibling_bwd_scan.py"; do
  echo "## python $cmd" >> $G
  timeout 900 python $cmd 2>&1 | grep -v amdgpu.ids | tail -12 >> $G
done
cat $G
