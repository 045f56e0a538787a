#!/bin/bash
# round 2, GPU call 1: correctness of the two-pass accumulation + gate calibration + first bench with extras
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2c1
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -s -x --deselect tests/test_bpr_gate_gpu.py -p no:cacheprovider > gpurun_out/r2c1/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2c1/pytest.log
timeout 600 python -m pytest tests/test_bpr_gate_gpu.py -m gpu -q -s -p no:cacheprovider > gpurun_out/r2c1/gate.log 2>&1
echo "gate rc=$?" >> gpurun_out/r2c1/gate.log
timeout 600 python bench.py > gpurun_out/r2c1/bench.json 2> gpurun_out/r2c1/bench.err
echo "bench rc=$?" >> gpurun_out/r2c1/bench.err
timeout 300 python scripts/bench_extra.py bpr_adagrad > gpurun_out/r2c1/adagrad.log 2>&1
tail -5 gpurun_out/r2c1/pytest.log; tail -30 gpurun_out/r2c1/gate.log; cat gpurun_out/r2c1/bench.json | head -c 6000; tail -3 gpurun_out/r2c1/adagrad.log
