#!/bin/bash
# round 4, GPU call 1: first run of the producer / consumer ALS kernel (als_pc.hpp): the half-epoch parity tests, then the A/B timing.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c1; mkdir -p $O
timeout 900 python -m pytest tests/test_als_gpu.py -q -m gpu -x -k "half_epochs or empty_rows or two_rank or resident" -p no:cacheprovider > $O/als_tests.txt 2>&1; tail -15 $O/als_tests.txt | cut -c1-300
timeout 600 python scripts/als_pc_ab.py --ablate > $O/als_pc_ab.txt 2>&1; tail -12 $O/als_pc_ab.txt | cut -c1-300
