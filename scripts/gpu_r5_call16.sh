#!/bin/bash
# GPU call 16 (round 5): the wide split kernel -- occupancy apart from code generation ("als_debug" 1024: two blocks per CU with the 168-register binary),
# and issue priority by role (variant libraries built with -DBFH_WIDE_PRIO=1 / 2: producer above consumers / consumers above producer)
O=gpurun_out/r5c16; mkdir -p $O
timeout 300 python scripts/als_wide_probe.py 160 --grid 2>&1 | grep "^d=" | tee $O/probe.txt
cp buffalo_amd/libbuffalo_hip.so /tmp/keep.so
for v in 1 2; do
  cp buffalo_amd/libbuffalo_hip_prio$v.so buffalo_amd/libbuffalo_hip.so
  echo "prio variant $v" | tee -a $O/probe.txt
  timeout 200 python scripts/als_wide_probe.py 160 --split-only 2>&1 | grep "^d=" | tee -a $O/probe.txt
done
cp /tmp/keep.so buffalo_amd/libbuffalo_hip.so
