#!/bin/bash
# GPU call 19 (round 5): the fp32 wide form at d = 192 and 256 after the tile-start change (d = 160: 24.3 -> 20.0 ms)
O=gpurun_out/r5c19; mkdir -p $O
for d in 192 256; do timeout 300 python scripts/als_wide_probe.py $d --split-only 2>&1 | grep "^d=" | tee -a $O/probe.txt; done
