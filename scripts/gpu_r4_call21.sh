#!/bin/bash
# GPU call 21: the BPR tests + gate + N-rank tests with the bias weight on by default
mkdir -p gpurun_out/r4c21
timeout 1200 python -m pytest tests/test_bpr_gpu.py tests/test_bpr_gate_gpu.py tests/test_comm_ranks_gpu.py tests/test_comm_gpu.py tests/test_large_gpu.py -q -x -s -m gpu > gpurun_out/r4c21/pytest.log 2>&1
echo "pytest rc=$?"
grep -n "oracle-a\|oracle-b\|  hip  \|top-10 overlap\|passed\|failed\|Error" gpurun_out/r4c21/pytest.log | head -40
tail -5 gpurun_out/r4c21/pytest.log
