#!/bin/bash
# study: the walk's speed when the (uniform) negatives are folded into the first R rows of Q -- R x 512 B per XCD replica
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c16; mkdir -p $O
export TMPDIR=/tmp
for m in "none" "im_neg_limit=6820" "im_neg_limit=3410" "im_neg_limit=1705" "im_neg_limit=852" "im_neg_limit=3410 --mode im_p_nt=1" "im_neg_limit=1705 --mode xcd_hot_tau=0"; do
  if [ "$m" = none ]; then MM=""; else MM="--mode $m"; fi
  tag=$(echo "$m" | tr ' =' '__' | tr -d '-')
  timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 60 $MM > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$tag.json")); print("$m |", round(d["value"]/1e9,3), "G/s", round(d["ms_per_step"],3), "ms/epoch  kernel", round(d["roofline"]["kernel_ms"],3), "ms")
except Exception as e:
    print("$m FAILED", e)
PY
done
