#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c6; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -s --maxfail=10 -p no:cacheprovider --durations=15 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
grep -E "passed|failed|FAILED|rc=" $O/pytest.log | tail -12; tail -2 $O/smoke.log; head -c 900 $O/bench.json; echo; tail -3 $O/bench.err
