#!/bin/bash
# rocprofv3 passes of the bench command (kernel stats of the very run that prints the bench line; PMC passes separately, as the
# MI355X guide prescribes) -> gpurun_out/$PROF_DIR (default r6prof); summaries are then copied under profiles/ by hand (committed).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${PROF_DIR:-r6prof}
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
# (--skip-extra bpr_lr005: that extra launches the HEADLINE kernel at another learning rate -- 7.5 ms per launch instead of 4.3 -- and would mix into its average)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- $CMD --skip-extra bpr_lr005 > $O/bench_under_rocprof.json 2> $O/stats.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- $CMD --no-extra --steps 3 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- $CMD --no-extra --steps 3 --warmup 1 > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -o p -- $CMD --no-extra --steps 3 --warmup 1 > $O/pmc_tcc.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o p -- $CMD --steps 2 --warmup 1 > $O/pmc_mfma.log 2>&1
cd $R
find $O -name "*kernel_trace.csv" -size +4M -delete
python scripts/pmc_summary.py $O $O/pmc_latest.json
for f in $(find $O/stats -name "*kernel_stats.csv"); do head -25 $f; done
cat $O/bench_under_rocprof.json | head -c 1200
