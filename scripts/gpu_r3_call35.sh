#!/bin/bash
# round 3, GPU call 35: the split pass with its scale taken from max|Q| alone -- ALS tests (and the timing A/B).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c35; mkdir -p $O
timeout 400 python -m pytest tests/test_als_gpu.py -x -q -m gpu -p no:cacheprovider > $O/als_tests.txt 2>&1; tail -4 $O/als_tests.txt | cut -c1-250
AB_TIMING_ONLY=1 timeout 100 python scripts/als_split_ab.py > $O/als_split_ab.txt 2>&1; tail -2 $O/als_split_ab.txt
