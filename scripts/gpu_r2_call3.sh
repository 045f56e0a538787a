#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c3; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_bpr_gpu.py tests/test_comm_gpu.py tests/test_residency_gpu.py tests/test_als_gpu.py tests/test_cfr_gpu.py -m gpu -q -s --maxfail=10 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python scripts/shard_times.py > $O/shards.log 2>&1
timeout 300 python scripts/bench_extra.py bpr_adagrad bpr_pcie > $O/extra.log 2>&1
bash scripts/gpu_profile.sh > $O/profile.log 2>&1
grep -E "passed|failed|FAILED|rc=" $O/pytest.log | tail -12; grep -E "item-major replay" $O/pytest.log; grep -E "^ALS d=" $O/pytest.log | awk '{print $2,$3,$4,$5,$6,$7,$8,$9,$10,$11,$12,$13,$14,$15,$16,$17,$18}' | sort -t' ' -k1,1 | head -70; tail -12 $O/shards.log; tail -3 $O/extra.log; tail -40 $O/profile.log | cut -c1-260
