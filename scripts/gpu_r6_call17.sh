#!/bin/bash
# GPU call 17 (round 6): where does the 3.87-4.35 ms spread of the headline kernel come from?  processes x handles on one box
O=gpurun_out/r6c17; mkdir -p $O
for i in 1 2 3 4; do timeout 300 python scripts/r6_walk_variance.py 2>&1 | grep "^handle"; echo "-- process $i"; done | tee $O/variance.txt
KEEP=1 REPS=4 timeout 300 python scripts/r6_walk_variance.py 2>&1 | grep "^handle" | tee -a $O/variance.txt
