#!/bin/bash
# round 3, GPU call 26: split-f16 ALS row kernel with round-to-nearest pieces -- ALS parity tests, then scripts/als_split_ab.py.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c26; mkdir -p $O
timeout 600 python -m pytest tests/test_als_gpu.py -q -m gpu > $O/als_tests.txt 2>&1; tail -12 $O/als_tests.txt | cut -c1-250
timeout 500 python scripts/als_split_ab.py > $O/als_split_ab.txt 2>&1; tail -8 $O/als_split_ab.txt
