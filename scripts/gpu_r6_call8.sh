#!/bin/bash
# GPU call 8 (round 6): the split-f16 wide kernel above vdim 160 (T = 6, 7, 8: four products l l + l h + h l + h h): parity cases at d = 192 / 224 / 256,
# then the ML-20M timings against the fp32 form
O=gpurun_out/r6c8; mkdir -p $O
timeout 1500 python -m pytest tests/test_als_gpu.py -q -m gpu -k "test_half_epochs_match_oracle and (192 or 224 or 256)" -s > $O/wide_parity.txt 2>&1; echo "wide parity rc=$?"; tail -4 $O/wide_parity.txt
grep -E "^ALS d=(192|224|256)" $O/wide_parity.txt | cut -c1-220
for d in 192 224 256; do timeout 600 python scripts/als_wide_probe.py $d 2>&1 | grep "^d="; done | tee $O/wide_times.txt
