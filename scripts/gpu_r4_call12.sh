#!/bin/bash
# round 4, GPU call 12: WARP parity at ML-20M scale (item side inside the unit ball this time)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c12; mkdir -p $O
timeout 600 python -m pytest tests/test_warp_scale_gpu.py -q -m gpu -rP -p no:cacheprovider > $O/warp_scale.txt 2>&1; tail -3 $O/warp_scale.txt | cut -c1-300; grep -E "^WARP at|P rows of|^E  " $O/warp_scale.txt | cut -c1-500
