#!/bin/bash
# GPU call 9 (round 5): als_wide_kernel at d = 160 -- the pass against the row ends (scripts/als_wide_probe.py)
O=gpurun_out/r5c9; mkdir -p $O
timeout 600 python scripts/als_wide_probe.py 160 > $O/probe.txt 2>&1; echo "rc=$?"; grep "^d=" $O/probe.txt; tail -3 $O/probe.txt | grep -v "^d="
