#!/bin/bash
# GPU call 28 (round 6): the next step's metadata prepared behind this step's loads, A/B on one box (im_dual_wg = 50: the same kernel without it)
O=gpurun_out/r6c28; mkdir -p $O
timeout 900 python -m pytest tests/test_bpr_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.txt
for m in '{"im_dual_wg":50}' '{}' '{"im_dual_wg":50}' '{}' '{"im_dual_wg":50}' '{}'; do echo "-- MODES=$m"; MODES="$m" REPS=2 timeout 300 python scripts/r6_walk_variance.py 2>&1 | grep "^handle" | cut -c1-70; done | tee $O/ab.txt
