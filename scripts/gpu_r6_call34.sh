#!/bin/bash
# GPU call 34 (round 6): ONE merge segment per epoch at N = 1 (as every N > 1 call already runs) instead of two: statistics against the stored oracle pairs, and the time
O=gpurun_out/r6c34; mkdir -p $O
for c in bench refbench lr0.05; do echo "== case $c"; CASE=$c REPS=2 SETTINGS='[{}, {"xcd_sync_updates": 33554432}]' timeout 900 python scripts/gate_knob_study.py 2>&1 | grep "^oracle\|^{" | cut -c1-420; done | tee $O/one_segment.txt
