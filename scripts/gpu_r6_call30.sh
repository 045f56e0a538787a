#!/bin/bash
# GPU call 30 (round 6): WARP trial kernel with the positive's row and its first candidate's row fetched one positive ahead -- parity tests, then timings
O=gpurun_out/r6c30; mkdir -p $O
timeout 1500 python -m pytest tests/test_warp_gpu.py tests/test_warp_scale_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee $O/tests.txt
timeout 600 python scripts/run_warp.py shape=ml20m epochs=6 2>&1 | tail -8 | tee $O/ml20m.txt
timeout 1200 python scripts/run_warp.py shape=c5 epochs=6 2>&1 | tail -8 | tee $O/c5.txt
