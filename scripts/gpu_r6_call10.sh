#!/bin/bash
# GPU call 10 (round 6): the Gramian without atomics (slice partials + ordered reduction); SPPMI fetch through the pinned ring; their tests; ALS + SPPMI extras
O=gpurun_out/r6c10; mkdir -p $O
timeout 600 python scripts/als_gramian_probe.py > $O/gramian.txt 2>&1; echo "gramian rc=$?"; grep "^d=" $O/gramian.txt
timeout 1500 python -m pytest tests/test_als_gpu.py tests/test_cfr_gpu.py tests/test_eals_gpu.py tests/test_sppmi.py -q -x -m gpu -k "not half_epochs" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
timeout 900 python bench.py --steps 20 --no-cpu-baseline --only-extra sppmi_ml20m_stream_w5 --only-extra als_ml20m_d128 --only-extra als_ml20m_d160 > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("bench_extra.json"))
e=d.get("extra",{})
for k in ("als_ml20m_d128","als_ml20m_d160","sppmi_ml20m_stream_w5"):
    v=e.get(k,{})
    print(k, {kk:v.get(kk) for kk in ("epoch_ms","kernel_ms_per_epoch","device_ms","wall_ms","error") if kk in v})
print("headline", d.get("value"), d.get("ms_per_step"), d["roofline"].get("kernel_ms"))
PY
