#!/bin/bash
# GPU call 40 (round 6): does the walk's mode follow the hardware queue its stream lands on?  k HIP streams created and kept before the first handle
O=gpurun_out/r6c40; mkdir -p $O
for k in 0 1 2 3 4 5 0 1 2 3 4 5; do echo "-- DUMMY_STREAMS=$k"; DUMMY_STREAMS=$k REPS=3 timeout 300 python scripts/r6_walk_variance.py 2>&1 | grep "^handle\|rror" | cut -c1-64; done | tee $O/streams.txt
