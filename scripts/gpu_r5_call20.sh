#!/bin/bash
# GPU call 20 (round 5): the whole GPU suite once more on another box (flakiness check of the final code), then smoke
O=gpurun_out/r5c20; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -x --durations=5 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -9 $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt
