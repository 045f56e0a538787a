#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c19; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bpr_gpu.py tests/test_bpr_gate_gpu.py tests/test_residency_gpu.py tests/test_comm_gpu.py tests/test_front_gpu.py -m gpu -q -s --maxfail=30 -p no:cacheprovider --durations=5 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
for m in "none" "im_dual=0"; do
  if [ "$m" = none ]; then MM=""; else MM="--mode $m"; fi
  tag=$(echo "$m" | tr ' =' '__' | tr -d '-')
  timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 100 $MM > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - <<PY
import json
d=json.load(open("$O/bench_$tag.json")); print("$m |", round(d["value"]/1e9,3), "G/s", round(d["ms_per_step"],3), "ms/epoch  kernel", round(d["roofline"]["kernel_ms"],3), "ms frac", round(d["roofline"]["frac"],3), d["roofline"]["kernel"])
PY
done
timeout 300 python scripts/shard_times.py > $O/shards.log 2>&1
CASE=lr0.05 SETTINGS='[{}]' timeout 400 python scripts/gate_knob_study.py > $O/study_lr005.log 2>&1
grep -E "passed|failed|FAILED|rc=|overlap|  hip" $O/pytest.log | tail -14 | cut -c1-220
grep -E "^N=|^shards8" $O/shards.log | cut -c1-200; grep -E "^\{" $O/study_lr005.log | cut -c1-250
