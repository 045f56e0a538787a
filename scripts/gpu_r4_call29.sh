#!/bin/bash
# GPU call 29: the library as shipped (als_solo_kernel compiled in, opt-in): every ALS case incl. the 'solo' design, smoke, a short bench line
mkdir -p gpurun_out/r4c29
timeout 600 python -m pytest tests/test_als_gpu.py tests/test_cfr_gpu.py tests/test_errors_gpu.py -q -x -m gpu -p no:cacheprovider > gpurun_out/r4c29/pytest_als.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r4c29/pytest_als.log | cut -c1-200
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4c29/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r4c29/smoke.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4c29/bench_short.json 2> gpurun_out/r4c29/bench_short.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open("gpurun_out/r4c29/bench_short.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","steps")})
print({k:v for k,v in d["roofline"].items() if k.startswith("als_") or k in ("kernel_ms","frac")})
P
