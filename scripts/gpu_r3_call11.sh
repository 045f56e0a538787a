#!/bin/bash
# round 3, GPU call 11: negative incidence list sorted + gathered in pieces of 2^27 pairs: parity (forced small pieces), configs[4] epoch.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c11; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_warp_gpu.py tests/test_bpr_gpu.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|FAILED|rc=|Error" $O/pytest.log | tail -5
timeout 300 python scripts/run_warp.py shape=c5 epochs=4 > $O/warp_c5.txt 2>&1; grep run_warp $O/warp_c5.txt | cut -c1-200
