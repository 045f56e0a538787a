#!/bin/bash
# GPU call 1 (round 6): the lr-0.05 question, measurements (a) and (b) of the round-5 review.
#  (b) sum or max: the walk with its negatives' atomics masked, the same atomics on a second stream, alone and side by side (scripts/r6_lr005_corun.py)
#  (a) width or order: the reference path (threaded oracle) at 8 / 64 / 128 / 256 workers on this box's host cores, then the walk under
#      im_blocks x im_max_stale (scripts/r6_lr005_width.py) -- the reference benchmark's setting (lr 0.05 -> 0.0001, 10 iterations)
O=gpurun_out/r6c1; mkdir -p $O
nproc > $O/box.txt; rocm-smi --showclocks >> $O/box.txt 2>&1
timeout 600 python scripts/r6_lr005_corun.py > $O/corun.txt 2>&1; echo "corun rc=$?"; grep -v "^$" $O/corun.txt | cut -c1-300 | tail -22
CASE=refbench WORKERS=8,64,128,256 timeout 1200 python scripts/r6_lr005_width.py > $O/width_refbench.txt 2>&1; echo "width rc=$?"
grep -E "^oracle|^hip" $O/width_refbench.txt | cut -c1-330
