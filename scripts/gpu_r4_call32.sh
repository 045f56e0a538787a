#!/bin/bash
# GPU call 32 (last of the round): the final library -- unpin_host synchronises its stream first -- bench line, smoke, the error / residency tests
mkdir -p gpurun_out/r4c32
timeout 150 python -X faulthandler bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4c32/bench.json 2> gpurun_out/r4c32/bench.err; echo "bench rc=$?"
grep -i "fault\|abort" gpurun_out/r4c32/bench.err | head -3
python - <<'P'
import json
try:
    d=json.loads(open("gpurun_out/r4c32/bench.json").read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","ms_per_step")}, [k for k,v in d["extra"].items() if isinstance(v,dict) and "error" in v])
except Exception as e: print("no line",e)
P
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4c32/smoke.log 2>&1; echo "smoke rc=$?"
timeout 150 python -m pytest tests/test_errors_gpu.py tests/test_residency_gpu.py -q -x -m gpu -p no:cacheprovider > gpurun_out/r4c32/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/r4c32/pytest.log | cut -c1-150
