#!/bin/bash
# GPU call 3 (round 5): the copy-back through the library's own pinned ring (pin_host opt-in), freed-caller-array test, soak of handle life cycles
# and bench runs, the host-buffer epoch with and without registration.
O=gpurun_out/r5c3; mkdir -p $O
timeout 900 python -m pytest tests/test_errors_gpu.py tests/test_residency_gpu.py tests/test_front_gpu.py -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
timeout 300 python scripts/bench_extra.py bpr_pcie > $O/pcie.txt 2>&1; echo "pcie rc=$?"; grep bpr_pcie $O/pcie.txt
bash scripts/soak_bench.sh $O/soak 25 4 20 > $O/soak_stdout.txt 2>&1; echo "soak rc=$?"; tail -45 $O/soak_stdout.txt
