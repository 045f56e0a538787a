#!/bin/bash
# GPU call 28: als_solo_kernel -- which parity cases pass, each in its own process (the first contact aborted in the d = 96 tiny case)
mkdir -p gpurun_out/r4c28
for k in "solo-128-kw15-ml100k" "solo-128-kw18-heavy" "solo-128-kw20-outliers" "solo-128-kw21-scales" "solo-96-kw12-tiny"; do
  echo "=== $k"
  timeout 120 python -m pytest tests/test_als_gpu.py -q -x -m gpu -k "$k" -p no:cacheprovider -p no:faulthandler > gpurun_out/r4c28/$k.log 2>&1
  echo "rc=$?"
  grep -v "^  File\|site-packages\|dist-packages" gpurun_out/r4c28/$k.log | tail -6 | cut -c1-300
done
