#!/bin/bash
# GPU call 9 (round 6): als_gramian_kernel knobs; the two-rank bench test; d = 160..224 parity of both wide forms
O=gpurun_out/r6c9; mkdir -p $O
timeout 600 python scripts/als_gramian_probe.py > $O/gramian.txt 2>&1; echo "gramian rc=$?"; grep "^d=" $O/gramian.txt
timeout 1200 python -m pytest tests/test_bench_ranks_gpu.py -q -x -m gpu > $O/bench_ranks.txt 2>&1; echo "bench ranks rc=$?"; tail -5 $O/bench_ranks.txt | cut -c1-300
timeout 1500 python -m pytest tests/test_als_gpu.py -q -m gpu -k "test_half_epochs_match_oracle and (160 or 192 or 224)" > $O/wide_parity.txt 2>&1; echo "wide parity rc=$?"; tail -3 $O/wide_parity.txt
