#!/bin/bash
# GPU call 8 (round 5): every vdim 160 / 192 parity case of the wide kernel, split-f16 ("inreg") beside the fp32 instruction ("fp32"), no early exit.
O=gpurun_out/r5c8; mkdir -p $O
timeout 1200 python -m pytest tests/test_als_gpu.py -q -s -k "160 and half_epochs and inreg and (tiny or heavy or scales)" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
grep "^ALS d=1[69]" $O/pytest.txt | sed -e "s/{'optimizer': '//" -e "s/'}//" | awk '{print $2,$3,$4,$5,$6,$7,$8, $10, $12, "ratio", $14}'
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --only-extra als_ml20m_d160 > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
s=open("gpurun_out/r5c8/bench.out").read().strip().split("\n")
e=json.loads(s[0][len("BENCH_EXTRA "):])
print(json.dumps(e["extra"].get("als_ml20m_d160")))
PY
