#!/bin/bash
# GPU call 6 (round 5): text ingestion on the device (parity with sscanf / the reference's compiled builder), its bench extra, ALS extra after the Gramian unroll.
O=gpurun_out/r5c6; mkdir -p $O
timeout 600 python -m pytest tests/test_ingest_gpu.py -x -q > $O/pytest_ingest.txt 2>&1; echo "pytest ingest rc=$?"; tail -6 $O/pytest_ingest.txt
timeout 600 python bench.py --steps 20 --warmup 5 --only-extra text_to_csr_2m_lines --only-extra coo_to_csr_ml20m --only-extra als_ml20m_d128 > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
s=open("gpurun_out/r5c6/bench.out").read().strip().split("\n")
e=json.loads(s[0][len("BENCH_EXTRA "):])
for k in ("text_to_csr_2m_lines","coo_to_csr_ml20m"):
    print(k, json.dumps(e["extra"].get(k))[:1500])
a=e["extra"].get("als_ml20m_d128",{})
print({k:v for k,v in a.items() if k in ("epoch_ms","kernel_ms_per_epoch","gramian_ff_ms_per_epoch","parity","error")})
print(len(s[-1]))
PY
tail -3 $O/bench.err
