#!/bin/bash
# GPU call 29 (round 6): the next slice pair fetched from inside the walk of the current one -- parity tests, then against the generic instantiation (same change) and call 26-28's numbers
O=gpurun_out/r6c29; mkdir -p $O
timeout 900 python -m pytest tests/test_bpr_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.txt
for m in '{}' '{}' '{}'; do echo "-- MODES=$m"; MODES="$m" REPS=3 timeout 300 python scripts/r6_walk_variance.py 2>&1 | grep "^handle" | cut -c1-70; done | tee $O/ab.txt
