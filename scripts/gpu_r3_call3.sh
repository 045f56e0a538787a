#!/bin/bash
# round 3, GPU call 3: front-level trace of the als_manual_cg_d64 whole-run gap; the exchange code with N > 1 ranks on one GPU
# (shm test transport), plus the files whose code paths changed (exchange arm / empty chunks / rank-independent weights / full hash).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c3; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python scripts/als_cg_diag.py front > $O/als_front_trace.txt 2>&1; echo "trace rc=$?" >> $O/als_front_trace.txt
timeout 900 python -m pytest tests/test_comm_ranks_gpu.py -m gpu -q -p no:cacheprovider -x --durations=8 > $O/pytest_ranks.log 2>&1
echo "pytest rc=$?" >> $O/pytest_ranks.log
timeout 600 python -m pytest tests/test_comm_gpu.py tests/test_residency_gpu.py tests/test_errors_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest_changed.log 2>&1
echo "pytest rc=$?" >> $O/pytest_changed.log
cat $O/als_front_trace.txt | tail -40
tail -60 $O/pytest_ranks.log; tail -15 $O/pytest_changed.log
