#!/bin/bash
# GPU call 5 (round 5): the full GPU suite on the round's code so far (copy-back ring, two libraries, compiled binding, Gramian unroll), smoke, one
# self-launched two-rank bench on the test library.
O=gpurun_out/r5c5; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
BFH_DEVICE_OVERRIDE=0 BFH_COMM_TRANSPORT=shm timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 > $O/n2.out 2> $O/n2.err; echo "n2 rc=$?"; tail -1 $O/n2.out | head -c 600; echo; tail -2 $O/n2.err
