#!/bin/bash
# round 3, GPU call 6: the quarantined warm-epoch config-#3 parity test (first measurement), the re-read micro-benchmark behind the
# ALS decision, then the WHOLE -m gpu suite as the driver runs it (-x) to confirm the state that is committed.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c6; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu_unmeasured -q -s -p no:cacheprovider > $O/pytest_unmeasured.log 2>&1
echo "pytest rc=$?" >> $O/pytest_unmeasured.log; grep -E "config #3|passed|failed|Error|rc=" $O/pytest_unmeasured.log | tail -30
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/micro/als_reread.hip -o /tmp/als_reread 2>/dev/null && timeout 120 /tmp/als_reread > $O/als_reread.txt 2>&1; cat $O/als_reread.txt
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=6 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|FAILED|rc=" $O/pytest.log | tail -5
