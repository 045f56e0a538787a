#!/bin/bash
# round 3, GPU call 30: split-f16 kernel after the ticket batches / fp32 division -- A/B timing and the float64 comparison.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c30; mkdir -p $O
timeout 400 python scripts/als_split_ab.py > $O/als_split_ab.txt 2>&1; tail -6 $O/als_split_ab.txt
