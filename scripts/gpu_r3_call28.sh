#!/bin/bash
# round 3, GPU call 28: quick loop for the split-f16 kernel -- the half-epoch parity cases, then the A/B timing.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c28; mkdir -p $O
timeout 300 python -m pytest tests/test_als_gpu.py -q -m gpu -k "half_epochs" > $O/als_tests.txt 2>&1; tail -4 $O/als_tests.txt | cut -c1-250
AB_TIMING_ONLY=1 timeout 300 python scripts/als_split_ab.py > $O/als_split_ab.txt 2>&1; tail -4 $O/als_split_ab.txt
