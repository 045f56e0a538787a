#!/bin/bash
# round 3, GPU call 7: the residual-first iALS++ gradient (als_kernels.hpp): parity files that touch it, the quarantined
# warm-epoch config-#3 test, and the config-#3 epoch time.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c7; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_als_gpu.py tests/test_cfr_gpu.py tests/test_large_gpu.py -m gpu -q -s -p no:cacheprovider > $O/pytest_als.log 2>&1
echo "pytest rc=$?" >> $O/pytest_als.log; grep -E "ratio|passed|failed|FAILED|rc=|Error" $O/pytest_als.log | tail -70
timeout 600 python -m pytest tests -m gpu_unmeasured -q -s -p no:cacheprovider > $O/pytest_unmeasured.log 2>&1
echo "pytest rc=$?" >> $O/pytest_unmeasured.log; grep -E "config #3|passed|failed|Error|rc=" $O/pytest_unmeasured.log | tail -30
timeout 300 python -c "
import bench, json
csr = bench.load_matrix('ml20m', 7)
e = bench.extra_als(csr, 7, cpu=False)
print(json.dumps({k: v for k, v in e.items() if k != 'epochs'}))
" > $O/als_extra.json 2>&1; tail -3 $O/als_extra.json | cut -c1-1500
