#!/bin/bash
# round 3, GPU call 1: what round 2 committed after its GPU budget was spent, first -- then the whole suite.
#   * sppmi.hip leaves out the end-of-file group (the reference's own compiled builder does: oracle/_ref)       -> tests/test_sppmi.py, test_cfr_gpu.py
#   * the stand-in fronts were rebuilt on shared helpers; CFR / EALS fronts train over the HIP backend for the first time -> tests/test_front_gpu.py
#   * stand-in loaders over the device builders against reference-built databases                                -> tests/test_data_loaders_ref.py
#   * whole training runs against stock-buffalo-over-oracle fixtures (tolerances set blind: tighten from this output) -> tests/test_trained_models_ref.py
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_r3_call1.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c1; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_sppmi.py tests/test_front_gpu.py tests/test_cfr_gpu.py tests/test_data_loaders_ref.py tests/test_trained_models_ref.py -m gpu -q -p no:cacheprovider > $O/pytest_unvalidated.log 2>&1
echo "pytest(unvalidated) rc=$?" >> $O/pytest_unvalidated.log
tail -15 $O/pytest_unvalidated.log
timeout 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
grep -E "passed|failed|FAILED|rc=|Fatal" $O/pytest.log | tail -8; tail -2 $O/smoke.log
