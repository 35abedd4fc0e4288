#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "39 64 32 2.0 fresh" "39 64 32 2.0 stress" "39 64 32 1.7 stress" "48 64 32 2.0 fresh" "39 48 32 2.0 fresh" "39 128 32 2.0 fresh" "39 96 32 2.0 fresh" "22 64 32 2.0 fresh" "39 64 128 2.0 fresh"; do
  set -- $cfg
  python tools/kbench.py --F $1 --E $2 --O $3 --alpha $4 --regime $5 2>&1 | tail -1
done
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_headline.py -x -q -m gpu -k "sweep or e64 or nfield or table_larger or g4 or g13" 2>&1 | tail -3
timeout 600 python tools/shape_scan.py --wide 2>&1 | tail -2
