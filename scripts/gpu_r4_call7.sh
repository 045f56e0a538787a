#!/bin/bash
# round 4, GPU call 7: scripts/micro/simd_overlap.hip (VALU wave + matrix wave on one SIMD), A/B timing of the ALS kernels
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c7; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/micro/simd_overlap.hip -o /tmp/simd_overlap 2>/dev/null && timeout 120 /tmp/simd_overlap > $O/simd_overlap.txt 2>&1; cat $O/simd_overlap.txt
timeout 600 python scripts/als_pc_ab.py --ablate --timing-only > $O/als_pc_ab.txt 2>&1; tail -7 $O/als_pc_ab.txt | cut -c1-300
