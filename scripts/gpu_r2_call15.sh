#!/bin/bash
# study: does the walk speed up when the negatives' rows fit the XCD's L2 (small catalogue), and does the nt hint on P help?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c15; mkdir -p $O
export TMPDIR=/tmp
for cfg in "ml20m none" "ml20m im_p_nt=1" "ml20m_i3410 none" "ml20m_i3410 im_p_nt=1" "ml20m_i3410 xcd_hot_tau=0" ; do
  set -- $cfg; shape=$1; m=$2
  if [ $m = none ]; then MM=""; else MM="--mode $m"; fi
  timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 60 --shape $shape $MM > $O/bench_${shape}_$m.json 2> $O/bench_${shape}_$m.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_${shape}_$m.json")); print("$shape $m", round(d["value"]/1e9,3), "G/s", round(d["ms_per_step"],3), "ms/epoch  kernel", round(d["roofline"]["kernel_ms"],3), "ms x", d["roofline"]["launches_per_step"])
except Exception as e:
    print("$shape $m FAILED", e)
PY
done
