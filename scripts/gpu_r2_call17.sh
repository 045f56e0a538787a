#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c17; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sppmi.py tests/test_cfr_gpu.py -m gpu -q -s --maxfail=20 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
cat > /tmp/sppmi_once.py <<PY
import sys, numpy as np
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bench
print(bench.extra_sppmi(bench.load_matrix("ml20m", 7), 7, cpu=False))
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o sppmi -- python /tmp/sppmi_once.py > $GRAFT_REPO_ROOT/$O/sppmi_prof.log 2>&1)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/sppmi_kernel_stats.csv; rm -rf $O/prof
grep -E "passed|failed|FAILED|rc=|sppmi of" $O/pytest.log | tail; grep -E "^\{" $O/sppmi_prof.log | cut -c1-900; head -12 $O/sppmi_kernel_stats.csv | cut -c1-170
