#!/bin/bash
# round 4, GPU call 10: producer with the transposing lane reduction + premultiplied row offsets; wide kernel (residual-first) at the 2.5x envelope
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c10; mkdir -p $O
timeout 900 python -m pytest tests/test_als_gpu.py -q -m gpu -rP -k "half_epochs or empty_rows or two_rank or resident" -p no:cacheprovider > $O/als_tests.txt 2>&1; tail -4 $O/als_tests.txt | cut -c1-300
grep -E "^ALS d=(160|192|224|256)" $O/als_tests.txt | cut -c1-190
grep -E "^FAILED|^ERROR" $O/als_tests.txt | head
timeout 600 python scripts/als_pc_ab.py --timing-only > $O/als_pc_ab.txt 2>&1; tail -3 $O/als_pc_ab.txt | cut -c1-300
