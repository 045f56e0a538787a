#!/bin/bash
# GPU call 14 (round 5): the wide split kernel at three blocks per CU (168 registers: the loops stay clean, the row ends spill) -- probe + parity at d = 160
O=gpurun_out/r5c14; mkdir -p $O
timeout 300 python scripts/als_wide_probe.py 160 2>&1 | grep "^d=" | tee $O/probe.txt
timeout 600 python -m pytest tests/test_als_gpu.py -q -k "160 and half_epochs and inreg" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
