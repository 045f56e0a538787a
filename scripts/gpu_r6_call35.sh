#!/bin/bash
# GPU call 35 (round 6): the per-handle draw against the size of the walk's working set (the item table, hence its 8 replicas, 1 / 2 / 4 times smaller)
O=gpurun_out/r6c35; mkdir -p $O
for d in 1 2 4 1 2 4; do echo "-- ITEM_DIV=$d"; ITEM_DIV=$d REPS=4 timeout 300 python scripts/r6_walk_variance.py 2>&1 | grep "^handle" | cut -c1-64; done | tee $O/item_div.txt
