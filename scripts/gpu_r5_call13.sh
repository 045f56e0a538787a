#!/bin/bash
# GPU call 13 (round 5): the wide split kernel with the residual on the last consumer -- parity cases at d = 160, the probe, the refbench gate case
O=gpurun_out/r5c13; mkdir -p $O
timeout 900 python -m pytest tests/test_als_gpu.py -q -s -k "160 and half_epochs and inreg" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
grep "^ALS d=" $O/pytest.txt | sed -e "s/{'optimizer': '//" -e "s/'}//" | awk '{print $2,$3,$4,$5,$6,$7,$8, $10, $12, "ratio", $14}' | grep -v block_size


