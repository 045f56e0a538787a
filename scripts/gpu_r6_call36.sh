#!/bin/bash
# GPU call 36 (round 6): als_gramian_kernel at 4 waves per CU for vdim 128 only -- the ALS / CFR / eALS parity files, then the ALS extra's epoch time
O=gpurun_out/r6c36; mkdir -p $O
timeout 1500 python -m pytest tests/test_als_gpu.py tests/test_cfr_gpu.py tests/test_eals_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed" | tee $O/tests.txt
timeout 300 python scripts/als_extra_only.py 2>&1 | tail -1 | tee $O/als_extra.txt
timeout 300 python scripts/als_extra_only.py 2>&1 | tail -1 | tee -a $O/als_extra.txt
