#!/bin/bash
# GPU call 11 (round 5): the wide kernel's CG quotients in fp32 (same bits as the double quotient rounded to float) -- wide parity cases + probes
O=gpurun_out/r5c11; mkdir -p $O
timeout 1200 python -m pytest tests/test_als_gpu.py -q -s -k "(160 or 192 or 224 or 256) and half_epochs and inreg" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
grep "^ALS d=" $O/pytest.txt | sed -e "s/{'optimizer': '//" -e "s/'}//" | awk '{print $2,$3,$4,$5,$6,$7,$8, $10, $12, "ratio", $14}' > $O/ratios.txt; awk '$NF > 2.4' $O/ratios.txt
timeout 600 python scripts/als_wide_probe.py 160 > $O/probe.txt 2>&1; grep "^d=" $O/probe.txt
timeout 600 python scripts/als_wide_probe.py 256 2>&1 | grep "^d=" | grep -v "no pass\|split  "
