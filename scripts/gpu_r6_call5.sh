#!/bin/bash
# GPU call 5 (round 6): als_ts_kernel with compile-time tile indices: A/B + ablations, then its parity cases
O=gpurun_out/r6c5; mkdir -p $O
timeout 900 python scripts/als_ts_ab.py --ablate > $O/ts_ab.txt 2>&1; echo "ab rc=$?"; grep -v "^$" $O/ts_ab.txt | cut -c1-200 | tail -20
timeout 900 python -m pytest tests/test_als_gpu.py -q -x -m gpu -k "test_half_epochs_match_oracle and ts and 128" -s > $O/ts_parity.txt 2>&1; echo "ts parity rc=$?"; tail -3 $O/ts_parity.txt
grep -E "^ALS d=128.*ts " $O/ts_parity.txt | cut -c1-200 | head -40
