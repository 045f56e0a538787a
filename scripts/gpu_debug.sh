export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_als_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_als.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_als.log
grep -E "^E  +Assert|^E  +assert|FAILED|passed|failed" gpurun_out/pytest_als.log | cut -c1-300 | tail -30
for f in -1 0 1; do echo "ALS_FUSED=$f"; ALS_FUSED=$f timeout 300 python scripts/bench_extra.py als 2>&1 | grep "^als" | cut -c1-120; done
echo "ALS_FUSED=1 ALS_DEBUG=1"; ALS_FUSED=1 ALS_DEBUG=1 timeout 300 python scripts/bench_extra.py als 2>&1 | grep "^als" | cut -c1-120
