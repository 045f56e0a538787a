#!/bin/bash
# round 3, GPU call 15: quarantined residency test (one changed key), BPRMF adagrad / host-buffer call pattern timings after the
# fused gather list and the full-buffer hash.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c15; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu_unmeasured -q -s -p no:cacheprovider > $O/pytest_unmeasured.log 2>&1
echo "pytest rc=$?" >> $O/pytest_unmeasured.log; grep -E "passed|failed|skipped|Error|rc=" $O/pytest_unmeasured.log | tail -5
timeout 400 python scripts/bench_extra.py bpr_adagrad bpr_pcie > $O/bench_extra.txt 2>&1; grep -E "^bpr_" $O/bench_extra.txt | cut -c1-400
