#!/bin/bash
# GPU call 18 (round 6): the headline kernel epoch by epoch from a cold process start (clock ramp?)
O=gpurun_out/r6c18; mkdir -p $O
for i in 1 2; do timeout 300 python scripts/r6_walk_ramp.py 2>&1 | grep -E "^cold|^after"; done | tee $O/ramp.txt
