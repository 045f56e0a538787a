#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c8; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_topk_gpu.py tests/test_sppmi.py tests/test_eals_gpu.py -m gpu -q -s --maxfail=60 -p no:cacheprovider --durations=8 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 400 python scripts/bench_extra.py topk > $O/topk_bench.log 2>&1; echo "topk bench rc=$?" >> $O/topk_bench.log
cp gpurun_out/bench_extra.json $O/bench_extra_topk.json 2>/dev/null
timeout 500 python scripts/gate_knob_study.py > $O/gate_knob_study.log 2>&1; echo "study rc=$?" >> $O/gate_knob_study.log
grep -E "passed|failed|FAILED|rc=|rows handed|sppmi of" $O/pytest.log | tail -40; grep -E "^topk|rc=|Error|error" $O/topk_bench.log | cut -c1-420 | tail; tail -40 $O/gate_knob_study.log | cut -c1-260
