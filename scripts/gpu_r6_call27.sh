#!/bin/bash
# GPU call 27 (round 6): the diet's second step (PLAIN instantiation, constant vdim) and 6 workgroups per CU, A/B on one box
O=gpurun_out/r6c27; mkdir -p $O
timeout 900 python -m pytest tests/test_bpr_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.txt
for m in '{"im_dual_generic":1}' '{}' '{"im_dual_wg":6}' '{"im_dual_generic":1}' '{}' '{"im_dual_wg":6}'; do echo "-- MODES=$m"; MODES="$m" REPS=2 timeout 300 python scripts/r6_walk_variance.py 2>&1 | grep "^handle" | cut -c1-70; done | tee $O/ab.txt
