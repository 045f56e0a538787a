export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_topk_gpu.py tests/test_errors_gpu.py tests/test_front_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_topk.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_topk.log
grep -E "^E  +|FAILED|passed|failed|rc=" gpurun_out/pytest_topk.log | cut -c1-300 | tail -30
timeout 600 python scripts/bench_extra.py topk 2>&1 | grep "^topk" | cut -c1-300
