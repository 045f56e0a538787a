#!/bin/bash
# GPU call 12 (round 6): the block solve's sums with the product rounded before the butterfly (as the select made it): every ALS / CFR / eALS test again
O=gpurun_out/r6c12; mkdir -p $O
timeout 2400 python -m pytest tests/test_als_gpu.py tests/test_cfr_gpu.py tests/test_eals_gpu.py -q -m gpu -s > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
grep -E "^ALS d=(96|128).*(tiny/inreg|outliers/fp32)" $O/tests.txt | cut -c1-200
timeout 600 python scripts/als_clock_probe.py > $O/clock.txt 2>&1; echo "clock rc=$?"; grep "user half" $O/clock.txt | cut -c1-220 | head -2
