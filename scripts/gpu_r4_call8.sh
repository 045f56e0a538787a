#!/bin/bash
# round 4, GPU call 8: simd_overlap micro with roles swapped / priorities; the ALS A/B once more on the restored placement
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c8; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/micro/simd_overlap.hip -o /tmp/simd_overlap 2>/dev/null && timeout 120 /tmp/simd_overlap > $O/simd_overlap.txt 2>&1; cat $O/simd_overlap.txt
timeout 600 python scripts/als_pc_ab.py --timing-only > $O/als_pc_ab.txt 2>&1; tail -3 $O/als_pc_ab.txt | cut -c1-300
