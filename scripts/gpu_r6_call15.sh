#!/bin/bash
# GPU call 15 (round 6): als_wide_kernel<SPLIT> at T = 7, 8 with the products issued block by block: times at d = 224 / 256 (split forced to T = 8), parity at 224 / 256
O=gpurun_out/r6c15; mkdir -p $O
for d in 224 256; do timeout 600 python - $d <<'PY' 2>&1 | grep "^d="
import sys, os
sys.argv=[sys.argv[0], sys.argv[1]]
src=open("scripts/als_wide_probe.py").read().replace('CASES = (("split", {}), ("split, no pass", {"als_debug": 16}), ("fp32", {"als_wide_split": 0}))','CASES = (("split", {"als_wide_split_max_t": 8}), ("split, no pass", {"als_wide_split_max_t": 8, "als_debug": 16}), ("fp32", {"als_wide_split": 0}))')
exec(compile(src, "scripts/als_wide_probe.py", "exec"))
PY
done | tee $O/wide_times.txt
timeout 1500 python -m pytest tests/test_als_gpu.py -q -m gpu -k "test_half_epochs_match_oracle and (224 or 256)" -s > $O/wide_parity.txt 2>&1; echo "wide parity rc=$?"; tail -3 $O/wide_parity.txt
grep -E "^ALS d=(224|256)" $O/wide_parity.txt | cut -c1-200
