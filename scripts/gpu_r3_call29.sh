#!/bin/bash
# round 3, GPU call 29: ablation of the split-f16 row kernel (als_debug bits 1 no solve, 2 no FF / FF p0, 16 no matrix instructions,
# 32 no row loads, 64 no weighting / cutting) -- timing only.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c29; mkdir -p $O
timeout 600 python scripts/als_ablation.py 0 115 3 64 > $O/als_ablation.txt 2>&1; grep als_debug $O/als_ablation.txt
