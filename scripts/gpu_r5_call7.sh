#!/bin/bash
# GPU call 7 (round 5): als_wide_kernel with the split-f16 Gramian (vdim 160 / 192): parity cases, then the d = 160 bench extra with it on and off.
O=gpurun_out/r5c7; mkdir -p $O
timeout 900 python -m pytest tests/test_als_gpu.py -x -q -k "160 or 192" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
grep "ALS d=1[69]" $O/pytest.txt | awk '{print $2,$3,$4,$5,$6,$7,$8,$9,$10,$15,$16}' | head -60
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --only-extra als_ml20m_d160 > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
s=open("gpurun_out/r5c7/bench.out").read().strip().split("\n")
e=json.loads(s[0][len("BENCH_EXTRA "):])
print(json.dumps(e["extra"].get("als_ml20m_d160")))
PY
tail -3 $O/bench.err
