#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c20; mkdir -p $O
export TMPDIR=/tmp
for m in "none" "xcd_sync_updates=33554432" "xcd_sync_updates=4194304" "im_dual=0"; do
  if [ "$m" = none ]; then MM=""; else MM="--mode $m"; fi
  tag=$(echo "$m" | tr ' =' '__' | tr -d '-')
  timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 100 $MM > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - <<PY
import json
d=json.load(open("$O/bench_$tag.json")); print("$m |", round(d["value"]/1e9,3), "G/s", round(d["ms_per_step"],3), "ms/epoch  kernel", round(d["roofline"]["kernel_ms"],3), "ms x", d["roofline"]["launches_per_step"], "frac", round(d["roofline"]["frac"],3))
PY
done
CASE=bench SETTINGS='[{"xcd_sync_updates":33554432},{}]' timeout 300 python scripts/gate_knob_study.py > $O/study_bench.log 2>&1
grep -E "^oracle|^\{" $O/study_bench.log | cut -c1-250
