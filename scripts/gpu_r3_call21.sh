#!/bin/bash
# round 3, GPU call 21: eALS with the dimensions walked in blocks of 16 (one 64-byte gather per entry and block): parity, epoch times.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c21; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_eals_gpu.py tests/test_front_gpu.py tests/test_trained_models_ref.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|FAILED|rc=|Error|assert" $O/pytest.log | tail -8
timeout 400 python scripts/bench_extra.py eals > $O/bench_extra.txt 2>&1; grep -E "^eals" $O/bench_extra.txt | cut -c1-400
