#!/bin/bash
# full GPU suite + smoke + the default bench line + the rocprofv3 passes of the same command (scripts/gpu_profile.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c14; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
bash scripts/gpu_profile.sh > $O/profile.log 2>&1
cp -r gpurun_out/r2prof $O/ 2>/dev/null
grep -E "passed|failed|FAILED|rc=" $O/pytest.log | tail -12; tail -2 $O/smoke.log; head -c 700 $O/bench.json; echo; tail -2 $O/bench.err; tail -30 $O/profile.log | cut -c1-220
