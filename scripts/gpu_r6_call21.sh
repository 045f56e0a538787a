#!/bin/bash
# GPU call 21 (round 6): the headline kernel against the placement of its buffers inside one allocation -- alignment and a per-block skew
O=gpurun_out/r6c21; mkdir -p $O
run() { echo "-- $*"; env "$@" BFH_ARENA_MB=6000 REPS=3 timeout 300 python scripts/r6_walk_variance.py 2>&1 | grep "^handle"; }
{
run BFH_ARENA_ALIGN_KB=2048
run BFH_ARENA_ALIGN_KB=4
run BFH_ARENA_ALIGN_KB=64
run BFH_ARENA_ALIGN_KB=2048 BFH_ARENA_SKEW_B=4096
run BFH_ARENA_ALIGN_KB=2048 BFH_ARENA_SKEW_B=69632
run BFH_ARENA_ALIGN_KB=2048 BFH_ARENA_SKEW_B=256
run BFH_ARENA_ALIGN_KB=1048576
run BFH_ARENA_ALIGN_KB=32768
} | tee $O/placement.txt
