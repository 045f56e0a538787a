#!/bin/bash
# round 4, GPU call 11: every rank's shard timed (N = 2, 4, 8), the N-rank Hogwild / ring-order tests, WARP parity at ML-20M scale
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c11; mkdir -p $O
timeout 600 python scripts/shard_times.py > $O/shard_times.txt 2>&1; grep -E "^N=|max |users per" $O/shard_times.txt | cut -c1-260
timeout 900 python -m pytest tests/test_comm_ranks_gpu.py -q -m gpu -rP -k "hogwild or ring_order" -p no:cacheprovider > $O/comm_tests.txt 2>&1; tail -3 $O/comm_tests.txt | cut -c1-300; grep -E "^Hogwild walk|ranks, .*ring-order|^FAILED" $O/comm_tests.txt | cut -c1-200
timeout 600 python -m pytest tests/test_warp_scale_gpu.py -q -m gpu -rP -p no:cacheprovider > $O/warp_scale.txt 2>&1; tail -3 $O/warp_scale.txt | cut -c1-300; grep -E "^WARP at|P rows of" $O/warp_scale.txt | cut -c1-400
