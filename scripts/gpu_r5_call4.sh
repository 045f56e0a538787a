#!/bin/bash
# GPU call 4 (round 5): vdim 160 on the wave-per-row split-f16 kernel (T = 5) -- parity cases and the bench extra; freed-caller-array test; soak part A.
O=gpurun_out/r5c4; mkdir -p $O
timeout 900 python -m pytest tests/test_als_gpu.py -x -q -k "160" > $O/pytest_als160.txt 2>&1; echo "pytest als160 rc=$?"; tail -5 $O/pytest_als160.txt
grep "ALS d=160" $O/pytest_als160.txt | head -40
timeout 300 python -m pytest tests/test_errors_gpu.py -x -q > $O/pytest_err.txt 2>&1; echo "pytest errors rc=$?"; tail -3 $O/pytest_err.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --only-extra als_ml20m_d160 --only-extra als_ml20m_d128 > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
s=open("gpurun_out/r5c4/bench.out").read().strip().split("\n")
e=json.loads(s[0][len("BENCH_EXTRA "):])
print(json.dumps(e["extra"].get("als_ml20m_d160")))
print({k:v for k,v in e["extra"].get("als_ml20m_d128",{}).items() if k in ("epoch_ms","kernel_ms_per_epoch")})
PY
tail -3 $O/bench.err
bash scripts/soak_bench.sh $O/soak 25 4 0 > $O/soak_stdout.txt 2>&1; echo "soak rc=$?"; tail -12 $O/soak_stdout.txt
