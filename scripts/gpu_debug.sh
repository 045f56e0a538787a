export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_cfr_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_cfr.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_cfr.log
grep -E "^E  +|FAILED|passed|failed|rc=" gpurun_out/pytest_cfr.log | cut -c1-300 | tail -30
