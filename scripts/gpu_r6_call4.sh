#!/bin/bash
# GPU call 4 (round 6): first contact of als_ts_kernel (the tile split): parity cases, then the A/B against the pairs
O=gpurun_out/r6c4; mkdir -p $O
timeout 900 python -m pytest tests/test_als_gpu.py -q -x -m gpu -k "test_half_epochs_match_oracle and ts" -s > $O/ts_parity.txt 2>&1; echo "ts parity rc=$?"; tail -4 $O/ts_parity.txt
grep -E "^ALS d=" $O/ts_parity.txt | cut -c1-200 | head -40
timeout 900 python scripts/als_ts_ab.py --ablate > $O/ts_ab.txt 2>&1; echo "ab rc=$?"; grep -v "^$" $O/ts_ab.txt | cut -c1-250 | tail -24
