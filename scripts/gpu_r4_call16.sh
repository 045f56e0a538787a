#!/bin/bash
# round 4, GPU call 16: T = 4: the f16 cut by the consumer, two instructions per matrix-instruction slot
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c16; mkdir -p $O
timeout 600 python scripts/als_pc_ab.py --ablate --timing-only > $O/als_pc_ab.txt 2>&1; tail -7 $O/als_pc_ab.txt | cut -c1-300
timeout 900 python -m pytest tests/test_als_gpu.py -q -m gpu -x -k "half_epochs or empty_rows or two_rank or resident" -p no:cacheprovider > $O/als_tests.txt 2>&1; tail -4 $O/als_tests.txt | cut -c1-300
