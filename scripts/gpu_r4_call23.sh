#!/bin/bash
# GPU call 23: micro-benchmark for the next ALS design -- waves that carry their own matrix + VALU mix, one and two per SIMD
mkdir -p gpurun_out/r4c23
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/micro/simd_fused_pairs.hip -o /tmp/simd_fused_pairs 2>/dev/null
timeout 120 /tmp/simd_fused_pairs | tee gpurun_out/r4c23/simd_fused_pairs.txt
