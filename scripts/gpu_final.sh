# Final measurement pass of the round: smoke, bench (+cpu baseline), rocprofv3 stats + PMC passes of the same command, tests.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
timeout 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
cat $O/bench_n1.json | cut -c1-1500
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bpr -o bpr -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/prof_bpr.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bpr -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bpr -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -o bpr -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_tcc.log 2>&1
cd $R
python scripts/pmc_summary.py $O $O/pmc_latest.json
for f in $(find $O/prof_bpr -name "*kernel_stats.csv"); do cut -c1-200 $f | head -8; done
find $O -name "*kernel_trace.csv" -size +4M -delete
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 200 python scripts/bench_extra.py bpr_pcie bpr_adagrad > $O/bench_extra_bpr.log 2>&1; grep -E "^bpr" $O/bench_extra_bpr.log | cut -c1-300
if [ -n "$FULL_TESTS" ]; then
  timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
fi
du -sh $O
