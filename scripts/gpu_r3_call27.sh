#!/bin/bash
# round 3, GPU call 27: per-row errors of the split-f16 pass on the stretches the config-#3 test samples.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c27; mkdir -p $O
timeout 500 python scripts/als_split_rows.py > $O/als_split_rows.txt 2>&1; tail -8 $O/als_split_rows.txt | cut -c1-400
