#!/bin/bash
# GPU call 19 (round 6): first handle of a process vs later ones -- with a large block allocated, touched and freed first
O=gpurun_out/r6c19; mkdir -p $O
for pw in 0 8 0 8 2; do echo "-- PREWARM=$pw GB"; PREWARM=$pw REPS=3 timeout 300 python scripts/r6_walk_variance.py 2>&1 | grep "^handle"; done | tee $O/variance.txt
