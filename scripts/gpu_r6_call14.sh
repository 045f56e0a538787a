#!/bin/bash
# GPU call 14 (round 6): is the d = 160 / block_size 64 tiny-case failure of call 13 deterministic?  Three runs of the case, then every ALS / CFR / eALS / front / ranks test
O=gpurun_out/r6c14; mkdir -p $O
for i in 1 2 3; do timeout 300 python -m pytest tests/test_als_gpu.py -q -m gpu -k "test_half_epochs_match_oracle and inreg-160-kw7-tiny" -s 2>&1 | grep -E "^ALS d=160|passed|failed" | cut -c1-200; done
timeout 2400 python -m pytest tests/test_als_gpu.py tests/test_cfr_gpu.py tests/test_eals_gpu.py tests/test_front_gpu.py tests/test_comm_ranks_gpu.py -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
