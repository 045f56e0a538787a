#!/bin/bash
# GPU call 33 (round 6): the two-triples walk as the default from 1024 users per queue (4- and 8-GPU shards): tests that run shards / ranks, then the per-rank times by default
O=gpurun_out/r6c33; mkdir -p $O
timeout 1500 python -m pytest tests/test_bpr_gpu.py tests/test_comm_ranks_gpu.py tests/test_bench_ranks_gpu.py tests/test_large_gpu.py tests/test_comm_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee $O/tests.txt
timeout 600 python scripts/shard_times.py 2>&1 | grep "^N=\|max " | cut -c1-330 | tee $O/shards_default.txt
