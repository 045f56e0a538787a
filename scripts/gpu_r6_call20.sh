#!/bin/bash
# GPU call 20 (round 6): the per-handle draw of the headline kernel with every device buffer cut from ONE allocation (BFH_ARENA_MB)
O=gpurun_out/r6c20; mkdir -p $O
for a in 0 6000 0 6000 6000 6000; do echo "-- BFH_ARENA_MB=$a"; if [ $a = 0 ]; then REPS=3 timeout 300 python scripts/r6_walk_variance.py 2>&1 | grep "^handle"; else BFH_ARENA_MB=$a REPS=3 timeout 300 python scripts/r6_walk_variance.py 2>&1 | grep "^handle"; fi; done | tee $O/variance.txt
