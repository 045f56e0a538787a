#!/bin/bash
# round 4, GPU call 6: do a VALU wave and a matrix wave on one SIMD overlap? (als_debug 512), later slot release
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c6; mkdir -p $O
timeout 600 python scripts/als_pc_ab.py --ablate --timing-only > $O/als_pc_ab.txt 2>&1; tail -9 $O/als_pc_ab.txt | cut -c1-300
