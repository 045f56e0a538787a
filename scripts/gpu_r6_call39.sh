#!/bin/bash
# GPU call 39 (round 6, STUDY build): virtual-memory-API buffers with the first GB of the process left unused; only buffers >= 8 MB through that path
O=gpurun_out/r6c39; mkdir -p $O
run() { echo "-- $*"; env "$@" REPS=3 timeout 300 python scripts/r6_walk_variance.py 2>&1 | grep "^handle\|rror" | cut -c1-64; }
{
run BFH_VMM_ALIGN_MB=2 BFH_VMM_BURN_MB=1024
run BFH_VMM_ALIGN_MB=2 BFH_VMM_BURN_MB=1024 BFH_VMM_MIN_MB=8
run BFH_VMM_ALIGN_MB=2 BFH_VMM_BURN_MB=4096
run BFH_VMM_ALIGN_MB=2
run BFH_VMM_ALIGN_MB=2 BFH_VMM_BURN_MB=1024
run BFH_VMM_ALIGN_MB=2 BFH_VMM_BURN_MB=1024 BFH_VMM_MIN_MB=8
run BFH_VMM_ALIGN_MB=2 BFH_VMM_BURN_MB=4096
run BFH_VMM_ALIGN_MB=2
run BFH_VMM_ALIGN_MB=0
} | tee $O/vmm_burn.txt
