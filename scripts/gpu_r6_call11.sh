#!/bin/bash
# GPU call 11 (round 6): the Gramian with the next trip's loads in flight; the block solve without its double conversions (bit-identical by construction):
# every ALS / CFR / eALS test, the row kernel's time
O=gpurun_out/r6c11; mkdir -p $O
timeout 600 python scripts/als_gramian_probe.py > $O/gramian.txt 2>&1; echo "gramian rc=$?"; grep "^d=" $O/gramian.txt
timeout 600 python scripts/als_clock_probe.py > $O/clock.txt 2>&1; echo "clock rc=$?"; grep "user half" $O/clock.txt | cut -c1-220
timeout 2400 python -m pytest tests/test_als_gpu.py tests/test_cfr_gpu.py tests/test_eals_gpu.py -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
