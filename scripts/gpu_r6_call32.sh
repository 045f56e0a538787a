#!/bin/bash
# GPU call 32 (round 6): after the diet, does the two-triples walk also win on the shards of a 4- and 8-GPU run (round 2: it lost there by 5 %)?
O=gpurun_out/r6c32; mkdir -p $O
for m in "" "im_dual=1" "" "im_dual=1"; do echo "== shard_times $m"; timeout 600 python scripts/shard_times.py $m 2>&1 | grep -v "^N=1\|users per rank" | cut -c1-330; done | tee $O/shards.txt
