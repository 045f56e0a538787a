export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_errors_gpu.py tests/test_als_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_err.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_err.log
grep -E "^E  +|FAILED|passed|failed|rc=" gpurun_out/pytest_err.log | cut -c1-300 | tail -40
timeout 300 python scripts/bench_extra.py als 2>&1 | grep "^als als_d128 " | cut -c1-200
