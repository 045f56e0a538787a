#!/bin/bash
# round 3, GPU call 13: the whole -m gpu suite as the driver runs it (fused gather lists for WARP and BPRMF adam/adagrad, residual-first
# ALS, everything since call 6), then the WARP profiler passes again on the final kernels (both shapes).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c13; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=6 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|FAILED|rc=" $O/pytest.log | tail -5
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
bash scripts/gpu_profile_warp.sh 2>&1 | grep -E "kernel|total|x " | tail -30
