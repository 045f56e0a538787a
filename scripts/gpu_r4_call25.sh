#!/bin/bash
# GPU call 25: the three SIMD micro-benchmarks again with -fno-slp-vectorize (the default build had paired the stand-in FMAs into v_pk_fma_f32:
# "200 FMAs" were 100 instructions), so that one FMA is one VALU instruction
mkdir -p gpurun_out/r4c25
for f in simd_overlap simd_fused_pairs simd_split_map; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize scripts/micro/$f.hip -o /tmp/$f 2>/dev/null
  echo "== $f (-fno-slp-vectorize)" | tee -a gpurun_out/r4c25/micro_noslp.txt
  timeout 120 /tmp/$f | tee -a gpurun_out/r4c25/micro_noslp.txt
done
