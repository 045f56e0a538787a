#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c5; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_als_gpu.py tests/test_comm_gpu.py -m gpu -q -s --maxfail=10 -p no:cacheprovider -k "spot or comm or topk" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python scripts/shard_times.py > $O/shards.log 2>&1
grep -E "passed|failed|FAILED|rc=" $O/pytest.log | tail -8; grep -E "config #3 spot|ALS d=128 iALS" $O/pytest.log | grep -v print; tail -12 $O/shards.log
