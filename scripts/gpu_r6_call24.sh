#!/bin/bash
# GPU call 24 (round 6): the per-handle draw with and without the next epoch's negatives drawn on the side stream while the walk runs
O=gpurun_out/r6c24; mkdir -p $O
for m in '{}' '{"im_presample_ahead":0}' '{}' '{"im_presample_ahead":0}' '{}' '{"im_presample_ahead":0}'; do echo "-- MODES=$m"; MODES="$m" REPS=4 timeout 300 python scripts/r6_walk_variance.py 2>&1 | grep "^handle"; done | tee $O/variance_presample.txt
