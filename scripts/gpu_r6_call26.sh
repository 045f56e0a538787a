#!/bin/bash
# GPU call 26 (round 6): the two-triples walk without per-lane guards / divergent readlanes / global sigmoid table -- parity tests, then A/B on one box
O=gpurun_out/r6c26; mkdir -p $O
timeout 1500 python -m pytest tests/test_bpr_gpu.py tests/test_bpr_gate_gpu.py -m gpu -x -q 2>&1 | tail -5 | tee $O/tests.txt
for m in '{"im_dual_generic":1}' '{}' '{"im_dual_generic":1}' '{}'; do echo "-- MODES=$m"; MODES="$m" REPS=3 timeout 300 python scripts/r6_walk_variance.py 2>&1 | grep "^handle" | cut -c1-70; done | tee $O/ab.txt
