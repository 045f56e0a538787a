#!/bin/bash
# GPU call 15 (round 5): where the pass of the wide split kernel goes at three blocks per CU -- "als_debug" 32 (consumers idle), 64 (producer idle)
O=gpurun_out/r5c15; mkdir -p $O
timeout 400 python scripts/als_wide_probe.py 160 --study 2>&1 | grep "^d=" | tee $O/probe.txt
