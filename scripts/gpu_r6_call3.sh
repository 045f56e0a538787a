#!/bin/bash
# GPU call 3 (round 6): the lr-aware stiffness rule of the per-XCD merge (xcd_stiff_lr_ref) on the reference benchmark's schedule and on the bench case
O=gpurun_out/r6c3; mkdir -p $O
SW='[{}, {"xcd_stiff_lr_ref": 0}, {"xcd_stiff_lr_ref": 1000}, {"xcd_stiff_lr_ref": 5000}, {"xcd_stiff_b": 0}]'
CASE=refbench WORKERS=8,16 SWEEP="$SW" timeout 900 python scripts/r6_lr005_width.py > $O/stiff_refbench.txt 2>&1; echo "refbench rc=$?"
grep -E "^oracle|^hip" $O/stiff_refbench.txt | cut -c1-330
CASE=bench WORKERS=8,64 SWEEP="$SW" timeout 900 python scripts/r6_lr005_width.py > $O/stiff_bench.txt 2>&1; echo "bench rc=$?"
grep -E "^oracle|^hip" $O/stiff_bench.txt | cut -c1-330
CASE=lr0.05 WORKERS= SWEEP='[{}]' timeout 600 python scripts/r6_lr005_width.py > $O/stiff_lr005.txt 2>&1; echo "lr0.05 rc=$?"
grep -E "^oracle|^hip" $O/stiff_lr005.txt | cut -c1-330
