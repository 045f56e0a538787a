#!/bin/bash
# GPU call 16 (round 6): times of als_wide_kernel<SPLIT> at d = 224 / 256 with the block-by-block products (ALS_WIDE_MAX_T=8: the split form at d = 256 too)
O=gpurun_out/r6c16; mkdir -p $O
for d in 224 256; do ALS_WIDE_MAX_T=8 timeout 600 python scripts/als_wide_probe.py $d 2>&1 | grep "^d="; done | tee $O/wide_times.txt
