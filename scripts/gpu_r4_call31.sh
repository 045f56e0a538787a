#!/bin/bash
# GPU call 31: the default build (als_solo_kernel compiled out): bench twice, smoke, the d = 128 ALS parity cases
mkdir -p gpurun_out/r4c31
for i in 1 2; do
  timeout 200 python -X faulthandler bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4c31/bench_$i.json 2> gpurun_out/r4c31/bench_$i.err; echo "bench $i rc=$?"
  grep -i "fault\|abort\|error" gpurun_out/r4c31/bench_$i.err | head -3
done
python - <<'P'
import json
for i in (1,2):
    try:
        d=json.loads(open("gpurun_out/r4c31/bench_%d.json"%i).read().strip().splitlines()[-1])
        print(i,{k:d[k] for k in ("value","ms_per_step")}, {k:v for k,v in d["roofline"].items() if k in ("kernel_ms","als_epoch_ms","als_kernel_ms","warp_ml20m_epoch_ms","warp_c5_epoch_ms")}, [k for k,v in d["extra"].items() if isinstance(v,dict) and "error" in v])
    except Exception as e: print(i,"no line",e)
P
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4c31/smoke.log 2>&1; echo "smoke rc=$?"
timeout 200 python -m pytest tests/test_als_gpu.py -q -x -m gpu -k "128 and (inreg or solo)" -p no:cacheprovider > gpurun_out/r4c31/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r4c31/pytest.log | cut -c1-200
