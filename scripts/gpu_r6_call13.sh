#!/bin/bash
# GPU call 13 (round 6): precompute without its blocking call (timers drained where the stream is idle next); ALS / CFR / eALS / dist / front tests; the ALS extras
O=gpurun_out/r6c13; mkdir -p $O
timeout 2400 python -m pytest tests/test_als_gpu.py tests/test_cfr_gpu.py tests/test_eals_gpu.py tests/test_front_gpu.py tests/test_comm_ranks_gpu.py -q -m gpu -x > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
timeout 900 python bench.py --steps 20 --no-cpu-baseline --only-extra als_ml20m_d128 --only-extra als_ml20m_d160 > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("bench_extra.json"))
e=d.get("extra",{})
for k in ("als_ml20m_d128","als_ml20m_d160"):
    v=e.get(k,{})
    print(k, {kk:v.get(kk) for kk in ("epoch_ms","kernel_ms_per_epoch","gramian_ff_ms_per_epoch","error") if kk in v})
PY
