export TMPDIR=/tmp
mkdir -p gpurun_out/prof_als
timeout 900 python -m pytest tests/test_als_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_als.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_als.log
grep -E "^E  +Assert|^E  +assert|FAILED|passed|failed" gpurun_out/pytest_als.log | cut -c1-300 | tail -30
cd /tmp
ALS_FUSED=${ALS_FUSED:-0} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_als -o als -- python $GRAFT_REPO_ROOT/scripts/bench_extra.py als 2>&1 | grep "^als" | cut -c1-150
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_als -name "*kernel_stats.csv" | head -1)
head -10 "$f" | cut -c1-200
