# Round-1 final measurement pass: tests, smoke, bench + rocprofv3 stats + PMC (separate passes), secondary benches.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bpr -o bpr -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/prof_bpr.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bpr -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bpr -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
cd $R
timeout 900 python scripts/bench_extra.py als warp bpr_adagrad bpr_pcie topk eals warp_c5 > $O/bench_extra.log 2>&1; cp gpurun_out/bench_extra.json $O/bench_extra.json
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_extra -o extra -- python $R/scripts/bench_extra.py als topk > $O/prof_extra.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_als -o als -- python $R/scripts/bench_extra.py als > $O/pmc_als.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_als_fetch -o als -- python $R/scripts/bench_extra.py als > $O/pmc_als_fetch.log 2>&1
cd $R
# keep only the summaries small enough to merge back
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +20M -exec sh -c 'head -200000 "$1" > "$1.head"; rm "$1"' _ {} \;
tail -2 $O/smoke.log; tail -3 $O/pytest.log; cat $O/bench_n1.json; grep -E "^(als|warp|bpr|topk)" $O/bench_extra.log | cut -c1-220
du -sh $O
