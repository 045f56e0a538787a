#!/bin/bash
# round 3, GPU call 33: the outliers case of the ALS half-epoch test on all three designs (no -x).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c33; mkdir -p $O
timeout 300 python -m pytest tests/test_als_gpu.py -q -m gpu -k "outliers" -s > $O/t.txt 2>&1; grep -E "outliers/|passed|failed" $O/t.txt | cut -c1-200
