export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_eals_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_eals.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_eals.log
grep -E "^E  +|FAILED|passed|failed|rc=" gpurun_out/pytest_eals.log | cut -c1-300 | tail -30
( time timeout 600 python scripts/bench_extra.py eals ) 2>&1 | grep -E "^eals|real|Error" | cut -c1-330
