#!/bin/bash
# round 3, GPU call 12: WARP gathers over the fused (user, coefficient) list: parity, then epoch times on both shapes.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c12; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_warp_gpu.py tests/test_comm_gpu.py tests/test_comm_ranks_gpu.py tests/test_trained_models_ref.py tests/test_front_gpu.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|FAILED|rc=|Error" $O/pytest.log | tail -5
timeout 200 python scripts/run_warp.py shape=ml20m epochs=4 > $O/warp_ml20m.txt 2>&1; grep run_warp $O/warp_ml20m.txt | cut -c1-200
timeout 300 python scripts/run_warp.py shape=c5 epochs=4 > $O/warp_c5.txt 2>&1; grep run_warp $O/warp_c5.txt | cut -c1-200
