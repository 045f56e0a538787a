#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c7; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_topk_gpu.py tests/test_sppmi.py tests/test_abi.py -m gpu -q -s --maxfail=20 -p no:cacheprovider --durations=8 > $O/pytest_topk_sppmi.log 2>&1
echo "pytest rc=$?" >> $O/pytest_topk_sppmi.log
timeout 400 python scripts/bench_extra.py topk > $O/topk_bench.log 2>&1; echo "topk bench rc=$?" >> $O/topk_bench.log
cp gpurun_out/bench_extra.json $O/bench_extra_topk.json 2>/dev/null
timeout 400 python -m pytest tests/test_bpr_gate_gpu.py -m gpu -q -s -p no:cacheprovider -k "lr0.05" > $O/pytest_gate.log 2>&1
echo "gate rc=$?" >> $O/pytest_gate.log
grep -E "passed|failed|FAILED|rc=|rows handed|sppmi of" $O/pytest_topk_sppmi.log | tail -30; grep -E "^topk|rc=|Error|error" $O/topk_bench.log | cut -c1-400 | tail; grep -E "oracle-|hip  |overlap|passed|failed|rc=|Assertion" $O/pytest_gate.log | cut -c1-300
