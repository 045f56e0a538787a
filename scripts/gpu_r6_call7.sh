#!/bin/bash
# GPU call 7 (round 6): power / clock of the box while the ALS row kernel runs; the spread-out matrix instructions probe (als_debug 2048);
# the configs[4]-shaped results check again (gradient rows)
O=gpurun_out/r6c7; mkdir -p $O
rocm-smi --showpower --showmaxpower --showperflevel > $O/power_idle.txt 2>&1; tail -12 $O/power_idle.txt | cut -c1-120
(for i in $(seq 1 40); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo; sleep 1; done > $O/power_during.txt) &
python - <<'PY' > $O/als_spread.txt 2>&1
import sys, os
sys.argv = ["x", "--timing-only"]
sys.path.insert(0, "scripts")
import numpy as np
exec(open("scripts/als_ts_ab.py").read().split('for m in ({"als_ts": 0}')[0])
for bits in (1024, 1024 + 2048, 1024, 1024 + 2048, 1024 + 16):
    timing({"als_ts": 0, "als_debug": bits}, epochs=6)
PY
echo "spread rc=$?"; grep -v "^$" $O/als_spread.txt | cut -c1-260 | tail -8
wait
sort $O/power_during.txt | uniq -c | sort -rn | head -12 | cut -c1-200
timeout 900 python -m pytest tests/test_warp_scale_gpu.py -q -x -m gpu -k tenth -s > $O/warp_tenth.txt 2>&1; echo "warp tenth rc=$?"; tail -3 $O/warp_tenth.txt; grep "WARP at" $O/warp_tenth.txt | cut -c1-1100
