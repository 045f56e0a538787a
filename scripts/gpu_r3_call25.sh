#!/bin/bash
# round 3, GPU call 25: the split-f16 ALS row kernel -- parity tests first, then the A/B (scripts/als_split_ab.py).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c25; mkdir -p $O
timeout 420 python -m pytest tests/test_als_gpu.py -x -q -m gpu > $O/als_tests.txt 2>&1; tail -5 $O/als_tests.txt
timeout 400 python scripts/als_split_ab.py > $O/als_split_ab.txt 2>&1; tail -8 $O/als_split_ab.txt
