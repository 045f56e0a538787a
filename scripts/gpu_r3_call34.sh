#!/bin/bash
# round 3, GPU call 34: rocprofv3 --kernel-trace --stats of the bench command on the final code (the ALS row kernel is the split-f16 one now).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c34
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- $CMD > $O/bench_under_rocprof.json 2> $O/stats.err
ls $O/stats; head -12 $O/stats/p_kernel_stats.csv | cut -c1-200
