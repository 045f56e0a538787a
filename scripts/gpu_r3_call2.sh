#!/bin/bash
# round 3, GPU call 2 (first executed): the ALS whole-run diagnostic (HIP / oracle / float64, pair by pair), then the whole
# -m gpu suite WITHOUT -x so one failure cannot hide the files behind it (round 2's WARP file never ran in the driver).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c2; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python scripts/als_cg_diag.py > $O/als_cg_diag.txt 2>&1; echo "diag rc=$?" >> $O/als_cg_diag.txt
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
cat $O/als_cg_diag.txt | tail -60
grep -E "passed|failed|FAILED|ERROR|rc=|Fatal" $O/pytest.log | tail -30; tail -2 $O/smoke.log
