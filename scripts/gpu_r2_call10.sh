#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c10; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bpr_gpu.py -m gpu -q -s --maxfail=30 -p no:cacheprovider --durations=8 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python scripts/shard_times.py im_user_replicas=0 > $O/shards_rep0.log 2>&1
timeout 300 python scripts/shard_times.py im_user_replicas=1 > $O/shards_rep1.log 2>&1
CASE=lr0.05 SETTINGS='[{"im_user_replicas":1},{"im_user_replicas":0}]' timeout 400 python scripts/gate_knob_study.py > $O/study_lr005.log 2>&1
CASE=bench SETTINGS='[{"im_user_replicas":1},{"im_user_replicas":0}]' timeout 300 python scripts/gate_knob_study.py > $O/study_bench.log 2>&1
grep -E "passed|failed|FAILED|rc=|ndcg" $O/pytest.log | tail -30
grep -E "^shards|^N=" $O/shards_rep0.log | cut -c1-230; echo; grep -E "^shards|^N=" $O/shards_rep1.log | cut -c1-230
grep -E "^oracle|^\{" $O/study_lr005.log | cut -c1-250; grep -E "^oracle|^\{" $O/study_bench.log | cut -c1-250
