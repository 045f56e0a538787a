export TMPDIR=/tmp
mkdir -p gpurun_out
free -g | head -2
( time timeout 900 python -m pytest tests/test_large_gpu.py -m gpu -q --timeout 800 -p no:cacheprovider -x ) > gpurun_out/pytest_large.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_large.log
grep -E "^E  +|FAILED|passed|failed|rc=|real" gpurun_out/pytest_large.log | cut -c1-300 | tail -20
timeout 300 python scripts/bench_extra.py als 2>&1 | grep "^als als_d128 " | cut -c1-200
