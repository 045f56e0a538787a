#!/bin/bash
# GPU call 18 (round 5): the three box-dependent gate cases (threaded oracle pair made on the box's CPU) once more on a fresh box, with their numbers,
# and the driver's exact bench command
O=gpurun_out/r5c18; mkdir -p $O
timeout 900 python -m pytest tests/test_bpr_gate_gpu.py -q -s -m gpu > $O/gate.txt 2>&1; echo "gate rc=$?"; tail -3 $O/gate.txt
grep -i -E "oracle|hip |norm|loss|prec|overlap" $O/gate.txt | cut -c1-230 | head -40
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.out | wc -c; tail -1 $O/bench.out | cut -c1-400
