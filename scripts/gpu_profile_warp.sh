#!/bin/bash
# rocprofv3 passes behind the WARP roofline: kernel durations (--kernel-trace --stats) and, in separate passes as the MI355X guide
# prescribes, FETCH_SIZE / WRITE_SIZE per kernel -- for bench.py's extra_warp workload (ML-20M shape, d=256) and for configs[4]'s
# shape on one GPU.  -> gpurun_out/warp_prof/{ml20m,c5}.json (+ kernel stats CSVs); copied under profiles/ by hand.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/warp_prof
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for SH in ml20m c5; do
  EP=4; [ $SH = c5 ] && EP=12
  CMD="python $R/scripts/run_warp.py shape=$SH epochs=$EP"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$SH/stats -o p -- $CMD out=$O/${SH}_epochs.json > $O/${SH}_stats.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/$SH/pmc_fetch -o p -- $CMD > $O/${SH}_fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/$SH/pmc_write -o p -- $CMD > $O/${SH}_write.log 2>&1
  python $R/scripts/pmc_kernels.py $O/$SH $EP $O/$SH.json > $O/${SH}_summary.txt 2>&1
  for f in $(find $O/$SH/stats -name "*kernel_stats.csv"); do cp $f $O/${SH}_kernel_stats.csv; done
  find $O/$SH -name "*.csv" -size +2M -delete
  grep run_warp $O/${SH}_stats.log | tail -4; cat $O/${SH}_summary.txt
done
