#!/bin/bash
# GPU call 24: micro-benchmark -- how a group's 30 matrix instructions and its VALU work can be dealt to the two waves of a SIMD
mkdir -p gpurun_out/r4c24
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/micro/simd_split_map.hip -o /tmp/simd_split_map 2>/dev/null
timeout 120 /tmp/simd_split_map | tee gpurun_out/r4c24/simd_split_map.txt
