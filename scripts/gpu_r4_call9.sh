#!/bin/bash
# round 4, GPU call 9: the f16 cut moved into the consumer's matrix-instruction shadow (als_pc.hpp); wide kernel on the residual-first gradient
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c9; mkdir -p $O
timeout 900 python -m pytest tests/test_als_gpu.py -q -m gpu -x -k "half_epochs or empty_rows or two_rank or resident" -p no:cacheprovider > $O/als_tests.txt 2>&1; tail -6 $O/als_tests.txt | cut -c1-300
grep -E "^ALS d=(160|192|224|256)" $O/als_tests.txt | cut -c1-190
timeout 600 python scripts/als_pc_ab.py --ablate --timing-only > $O/als_pc_ab.txt 2>&1; tail -7 $O/als_pc_ab.txt | cut -c1-300
