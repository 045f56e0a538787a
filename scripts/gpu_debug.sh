export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ingest_gpu.py tests/test_front_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -s > gpurun_out/pytest_ingest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_ingest.log
grep -E "^E  +|FAILED|passed|failed|rc=|Error|^ingest|ingest 20M" gpurun_out/pytest_ingest.log | cut -c1-300 | tail -40
