#!/bin/bash
# round 3, GPU call 10: WARP trial kernel with the next positive's rows one ahead + occupancy-derived persistent grid:
# parity (trial counts identical to the oracle), then epoch times on both shapes.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c10; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_warp_gpu.py tests/test_comm_gpu.py tests/test_comm_ranks_gpu.py tests/test_front_gpu.py tests/test_trained_models_ref.py -m gpu -q -p no:cacheprovider -x > $O/pytest_warp.log 2>&1
echo "pytest rc=$?" >> $O/pytest_warp.log; grep -E "passed|failed|FAILED|rc=|Error" $O/pytest_warp.log | tail -5
timeout 200 python scripts/run_warp.py shape=ml20m epochs=4 > $O/warp_ml20m.txt 2>&1; grep run_warp $O/warp_ml20m.txt | cut -c1-330
timeout 300 python scripts/run_warp.py shape=c5 epochs=6 > $O/warp_c5.txt 2>&1; grep run_warp $O/warp_c5.txt | cut -c1-330
