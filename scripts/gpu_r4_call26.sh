#!/bin/bash
# GPU call 26: what the memory system gives for 20 M random 512-byte rows (the ALS row kernel's gather), by table size, key order, waves per CU and groups in flight
mkdir -p gpurun_out/r4c26
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/micro/gather_rate.hip -o /tmp/gather_rate 2>/dev/null
timeout 200 /tmp/gather_rate | tee gpurun_out/r4c26/gather_rate.txt
