#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c9; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_topk_gpu.py -m gpu -q -s --maxfail=60 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 400 python scripts/bench_extra.py topk > $O/topk_bench.log 2>&1; echo "topk bench rc=$?" >> $O/topk_bench.log
cp gpurun_out/bench_extra.json $O/bench_extra_topk.json 2>/dev/null
for m in none im_max_stale=32 im_max_stale=16 im_max_stale=8; do
  if [ $m = none ]; then MM=""; else MM="--mode $m"; fi
  timeout 200 python bench.py --no-extra --no-cpu-baseline --steps 100 $MM > $O/bench_$m.json 2> $O/bench_$m.err
done
CASE=lr0.05 SETTINGS='[{"im_max_stale":32},{"im_max_stale":16},{"im_max_stale":8},{"im_max_stale":16,"im_blocks":16},{"im_max_stale":8,"im_blocks":16},{"im_max_stale":16,"im_drift_budget":500}]' timeout 400 python scripts/gate_knob_study.py > $O/study_lr005.log 2>&1
CASE=bench SETTINGS='[{},{"im_max_stale":16},{"im_max_stale":8}]' timeout 300 python scripts/gate_knob_study.py > $O/study_bench.log 2>&1
grep -E "passed|failed|FAILED|rc=" $O/pytest.log | tail -20; grep -E "^topk|rc=|Error|error" $O/topk_bench.log | cut -c1-420 | tail -6
for m in none im_max_stale=32 im_max_stale=16 im_max_stale=8; do python - <<PY
import json
d=json.load(open("$O/bench_$m.json")); print("$m", round(d["value"]/1e9,3), "G/s", round(d["ms_per_step"],3), "ms", "kernel", round(d["roofline"]["kernel_ms"],3), "frac", round(d["roofline"]["frac"],3))
PY
done
grep -E "^oracle|^\{" $O/study_lr005.log | cut -c1-250; grep -E "^oracle|^\{" $O/study_bench.log | cut -c1-250
