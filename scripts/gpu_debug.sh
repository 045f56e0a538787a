export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "^E  +Assert|^E  +assert|FAILED|passed|failed|rc=" gpurun_out/pytest_gpu.log | cut -c1-300 | tail -30
timeout 300 python scripts/bench_extra.py als 2>&1 | grep "^als" | cut -c1-330
