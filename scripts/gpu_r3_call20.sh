#!/bin/bash
# round 3, GPU call 20: per-rank cost of bench.py --gpus N on ONE GPU (rank 0's shard, one-rank communicator) with the heavy-user
# replicas -- the compute half of the 8-GPU projection.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c20; mkdir -p $O
timeout 600 python scripts/shard_times.py > $O/shard_times.txt 2>&1; grep -E "^shards|^N=" $O/shard_times.txt | cut -c1-220
