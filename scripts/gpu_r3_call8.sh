#!/bin/bash
# round 3, GPU call 8: residual-first gradient with the ds_swizzle half-sum: config-#3 epoch time, the warm-epoch test
# (item half-epoch also from the oracle's inputs), the ALS parity file.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c8; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "
import bench, json
csr = bench.load_matrix('ml20m', 7)
e = bench.extra_als(csr, 7, cpu=False)
print(json.dumps({k: v for k, v in e.items() if k != 'epochs'}))
" > $O/als_extra.json 2>&1; tail -1 $O/als_extra.json | cut -c1-700
timeout 600 python -m pytest tests -m gpu_unmeasured -q -s -p no:cacheprovider > $O/pytest_unmeasured.log 2>&1
echo "pytest rc=$?" >> $O/pytest_unmeasured.log; grep -E "config #3|passed|failed|Error|rc=" $O/pytest_unmeasured.log | tail -30
timeout 900 python -m pytest tests/test_als_gpu.py tests/test_cfr_gpu.py -m gpu -q -p no:cacheprovider -x > $O/pytest_als.log 2>&1
echo "pytest rc=$?" >> $O/pytest_als.log; grep -E "passed|failed|FAILED|rc=|Error" $O/pytest_als.log | tail -5
