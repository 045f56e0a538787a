#!/bin/bash
# round 3, GPU call 36: the other test files that train ALS-family models, after the split pass's scale rule changed.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c36; mkdir -p $O
timeout 150 python -m pytest tests/test_trained_models_ref.py tests/test_front_gpu.py tests/test_cfr_gpu.py tests/test_eals_gpu.py tests/test_large_gpu.py -x -q -m gpu -p no:cacheprovider -k "not bpr and not warp" > $O/t.txt 2>&1; tail -3 $O/t.txt | cut -c1-250
