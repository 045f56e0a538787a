#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c11; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bpr_gpu.py tests/test_bpr_gate_gpu.py -m gpu -q -s --maxfail=30 -p no:cacheprovider --durations=6 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python scripts/shard_times.py > $O/shards_auto.log 2>&1
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_topk -o topk -- python $GRAFT_REPO_ROOT/scripts/bench_extra.py topk > $GRAFT_REPO_ROOT/$O/topk_prof.log 2>&1)
find $O/prof_topk -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/topk_kernel_stats.csv
rm -rf $O/prof_topk
grep -E "passed|failed|FAILED|rc=|overlap|hip  " $O/pytest.log | tail -30 | cut -c1-250
grep -E "^shards|^N=" $O/shards_auto.log | cut -c1-230
head -14 $O/topk_kernel_stats.csv | cut -c1-200
