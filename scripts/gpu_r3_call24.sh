#!/bin/bash
# round 3, GPU call 24: ablation of the config-#3 ALS row kernel per half-epoch (scripts/als_ablation.py).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c24; mkdir -p $O
timeout 600 python scripts/als_ablation.py 0 1 2 4 3 7 > $O/als_ablation.txt 2>&1; grep als_debug $O/als_ablation.txt
