#!/bin/bash
# GPU call 23 (round 6): what differs between a slow and a fast HANDLE of the headline kernel?  Counter passes over four handles in one process each.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c23; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $O/counters_available.txt 2>&1
grep -c . $O/counters_available.txt
pass() { n=$1; shift; REPS=4 timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- python $R/scripts/r6_walk_variance.py > $O/$n.log 2>&1; echo "== $n: $*"; grep "^handle" $O/$n.log | cut -c1-60; python $R/scripts/r6_walk_pmc.py $O/$n; find $O/$n -name "*.csv" -size +1M -delete; }
{
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum
pass ea TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum
pass busy GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES
} 2>&1 | tee $O/summary.txt
