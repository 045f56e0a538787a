#!/bin/bash
# round 3, GPU call 9: config-#3 epoch time of the residual-first kernel with the compile-time LOSS switch (A, the committed
# library) and with the half-sum swizzles placed early in the interleave pattern (B, scripts/micro/variants); the warm-epoch test
# calibrated by the oracle's own reordering spread; ALS / CFR parity on A.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c9; mkdir -p $O
export TMPDIR=/tmp
T='
import bench, json, sys
csr = bench.load_matrix("ml20m", 7)
e = bench.extra_als(csr, 7, cpu=False)
print(sys.argv[1], json.dumps({k: e[k] for k in ("epoch_ms", "kernel_ms_per_epoch")}), e["mfma"]["frac"])
'
timeout 200 python -c "$T" A > $O/als_time.txt 2>&1
cp buffalo_amd/libbuffalo_hip.so /tmp/libA.so
cp scripts/micro/variants/libbuffalo_hip_vB.so buffalo_amd/libbuffalo_hip.so; timeout 200 python -c "$T" B >> $O/als_time.txt 2>&1
cp /tmp/libA.so buffalo_amd/libbuffalo_hip.so; timeout 200 python -c "$T" A2 >> $O/als_time.txt 2>&1
grep -E "^A|^B" $O/als_time.txt
timeout 600 python -m pytest tests -m gpu_unmeasured -q -s -p no:cacheprovider > $O/pytest_unmeasured.log 2>&1
echo "pytest rc=$?" >> $O/pytest_unmeasured.log; grep -E "config #3|top-10|passed|failed|Error|rc=" $O/pytest_unmeasured.log | grep -v print | tail -30
timeout 900 python -m pytest tests/test_als_gpu.py tests/test_cfr_gpu.py tests/test_front_gpu.py tests/test_trained_models_ref.py -m gpu -q -p no:cacheprovider -x > $O/pytest_als.log 2>&1
echo "pytest rc=$?" >> $O/pytest_als.log; grep -E "passed|failed|FAILED|rc=|Error" $O/pytest_als.log | tail -5
