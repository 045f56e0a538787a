#!/bin/bash
# GPU call 38 (round 6, STUDY build): device buffers through the virtual-memory API with the virtual address aligned to 2 MB / 1 GB, against hipMalloc -- the walk's per-handle draw
O=gpurun_out/r6c38; mkdir -p $O
for a in 0 1024 2 0 1024 2 0 1024 2 1024 1024; do echo "-- BFH_VMM_ALIGN_MB=$a"; BFH_VMM_ALIGN_MB=$a REPS=3 timeout 300 python scripts/r6_walk_variance.py 2>&1 | grep "^handle\|rror" | cut -c1-130; done | tee $O/vmm.txt
