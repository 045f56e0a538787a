#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c18; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bpr_gpu.py -m gpu -q -s --maxfail=30 -p no:cacheprovider -k "conflict_free or statistical_parity" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
for m in "none" "im_dual=1" "im_dual=1 --mode xcd_hot_tau=0" "im_dual=1 --mode waves_per_cu=16" "im_dual=1 --mode waves_per_cu=20"; do
  if [ "$m" = none ]; then MM=""; else MM="--mode $m"; fi
  tag=$(echo "$m" | tr ' =' '__' | tr -d '-')
  timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 60 $MM > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$tag.json")); print("$m |", round(d["value"]/1e9,3), "G/s", round(d["ms_per_step"],3), "ms/epoch  kernel", round(d["roofline"]["kernel_ms"],3), "ms", d["roofline"]["kernel"])
except Exception as e:
    print("$m FAILED", e); print(open("$O/bench_$tag.err").read()[-600:])
PY
done
CASE=bench SETTINGS='[{"im_dual":1}]' timeout 300 python scripts/gate_knob_study.py > $O/study_bench.log 2>&1
CASE=lr0.05 SETTINGS='[{"im_dual":1}]' timeout 400 python scripts/gate_knob_study.py > $O/study_lr005.log 2>&1
grep -E "passed|failed|FAILED|rc=|ndcg" $O/pytest.log | tail -14
grep -E "^oracle|^\{" $O/study_bench.log | cut -c1-250; grep -E "^oracle|^\{" $O/study_lr005.log | cut -c1-250
