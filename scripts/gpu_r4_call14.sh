#!/bin/bash
# round 4, GPU call 14: producer with keys two chunks ahead + L2 touch of the next chunk's rows
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c14; mkdir -p $O
timeout 600 python scripts/als_pc_ab.py --ablate --timing-only > $O/als_pc_ab.txt 2>&1; tail -7 $O/als_pc_ab.txt | cut -c1-300
timeout 900 python -m pytest tests/test_als_gpu.py -q -m gpu -x -k "half_epochs or empty_rows or two_rank or resident" -p no:cacheprovider > $O/als_tests.txt 2>&1; tail -4 $O/als_tests.txt | cut -c1-300
