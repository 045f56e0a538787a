#!/bin/bash
# GPU call 12 (round 5): the kernel-stats pass of the bench command again, without the lr-0.05 extra (its launches of the headline kernel had mixed into the
# average: 5.44 ms over 70 launches); the refbench gate case with the bounds of the GPU box's own oracle pair.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c12; mkdir -p $O
timeout 600 python -m pytest tests/test_bpr_gate_gpu.py -q -s -k refbench > $O/pytest_gate.txt 2>&1; echo "gate rc=$?"; grep "oracle-a\|oracle-b\|hip  \|overlap\|passed\|failed" $O/pytest_gate.txt | cut -c1-200
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --skip-extra bpr_lr005 > $O/bench_under_rocprof.json 2> $O/stats.err; echo "stats rc=$?"
cd $R
for f in $(find $O/stats -name "*kernel_stats.csv"); do cp $f $O/bench_rocprofv3_kernel_stats.csv; head -6 $f | cut -c1-150; done
find $O -name "*kernel_trace.csv" -size +4M -delete
tail -1 $O/bench_under_rocprof.json | cut -c1-700
