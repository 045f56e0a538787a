#!/bin/bash
# GPU call 41 (round 6): is the first handle of a process the slow one because it allocates while device code / sort storage / streams are first created?  A small throw-away handle first.
O=gpurun_out/r6c41; mkdir -p $O
for w in 0 1 2 0 1 2 0 1 2; do echo "-- WARM_TINY=$w"; WARM_TINY=$w REPS=3 timeout 300 python scripts/r6_walk_variance.py 2>&1 | grep "^handle\|^tiny\|rror" | cut -c1-64; done | tee $O/warm_tiny.txt
