#!/bin/bash
# GPU call 27: als_solo_kernel (als_pc = 3) -- first contact: the parity cases of the 'solo' design under a short timeout, then the A/B timing
mkdir -p gpurun_out/r4c27
timeout 300 python -m pytest tests/test_als_gpu.py -q -x -m gpu -k "solo" -p no:cacheprovider > gpurun_out/r4c27/pytest_solo.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r4c27/pytest_solo.log | cut -c1-220
timeout 240 python scripts/als_pc_ab.py > gpurun_out/r4c27/ab.txt 2>&1
echo "ab rc=$?"; tail -12 gpurun_out/r4c27/ab.txt | cut -c1-250
