#!/bin/bash
# GPU call 6 (round 6): (1) the shader clock the ALS row kernels see with and without their matrix instructions (als_debug bit 1024);
# (2) the results check at a tenth of configs[4]; (3) the three BPRMF gate cases with the lr-aware merge weight and the refbench case on the 128 / 256 pair
O=gpurun_out/r6c6; mkdir -p $O
timeout 600 python scripts/als_ts_ab.py --clock --timing-only > $O/ts_clock.txt 2>&1; echo "clock rc=$?"; grep -v "^$" $O/ts_clock.txt | cut -c1-260 | tail -14
timeout 900 python -m pytest tests/test_warp_scale_gpu.py -q -x -m gpu -k tenth -s > $O/warp_tenth.txt 2>&1; echo "warp tenth rc=$?"; tail -3 $O/warp_tenth.txt; grep "WARP at" $O/warp_tenth.txt | cut -c1-900
timeout 1500 python -m pytest tests/test_bpr_gate_gpu.py -q -s -m gpu > $O/gate.txt 2>&1; echo "gate rc=$?"; tail -3 $O/gate.txt
grep -i -E "oracle-|hip  |overlap" $O/gate.txt | cut -c1-230 | head -40
