#!/bin/bash
# GPU call 2 (round 6): (b) again with the HIP runtime loaded in the right order; the walk's bias vectors at refbench after 1 / 2 / 10 epochs (the oracle's
# were made in the build container: |Qb| of the reference path does not depend on the pool width); the two advisor fixes' tests.
O=gpurun_out/r6c2; mkdir -p $O
timeout 600 python scripts/r6_lr005_corun.py > $O/corun.txt 2>&1; echo "corun rc=$?"; grep "^{" $O/corun.txt | cut -c1-400
for e in 1 2 10; do EPOCHS=$e WHO=hip timeout 300 python scripts/r6_qb_dump.py 2>&1 | tail -1; done
EPOCHS=10 WHO=hip TAG=_atomic MODES='{"hogwild_atomic": 1}' timeout 300 python scripts/r6_qb_dump.py 2>&1 | tail -1
timeout 900 python -m pytest tests/test_ingest_gpu.py -q -x -m gpu -k "text or handed or plain" > $O/ingest.txt 2>&1; echo "ingest rc=$?"; tail -3 $O/ingest.txt
timeout 900 python -m pytest tests/test_als_gpu.py -q -x -m gpu -k "heavy_outliers or (160 and outliers)" > $O/als.txt 2>&1; echo "als rc=$?"; tail -3 $O/als.txt
