export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_als_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_als.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_als.log
grep -E "^E  +Assert|^E  +assert|FAILED|passed|failed" gpurun_out/pytest_als.log | cut -c1-300 | tail -30
timeout 300 python scripts/bench_extra.py als 2>&1 | grep "^als" | cut -c1-200
