#!/bin/bash
# round 4, GPU call 15: the pc kernel against the wave-per-row kernel at d = 64 and d = 96 (T = 2, 3)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c15; mkdir -p $O
for d in 64 96; do ALS_D=$d timeout 400 python scripts/als_pc_ab.py --timing-only > $O/als_pc_ab_d$d.txt 2>&1; echo "d=$d"; tail -2 $O/als_pc_ab_d$d.txt | cut -c1-200; done
