#!/bin/bash
# round 4, GPU call 3: where the producer's time goes (als_debug probes), the remaining ALS tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c3; mkdir -p $O
timeout 600 python scripts/als_pc_ab.py --ablate --timing-only > $O/als_pc_ab.txt 2>&1; tail -9 $O/als_pc_ab.txt | cut -c1-300
timeout 900 python -m pytest tests/test_als_gpu.py -q -m gpu -x -k "resident or scales or outliers" -p no:cacheprovider > $O/als_tests.txt 2>&1; tail -8 $O/als_tests.txt | cut -c1-300
grep -E "^ALS d=128.*(scales|outliers)" $O/als_tests.txt | cut -c1-200
