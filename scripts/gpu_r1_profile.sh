set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
timeout 120 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke_r1.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_r1.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_r1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_r1.log
timeout 400 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r1.json 2> gpurun_out/bench_r1.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1_bpr -o bpr -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_r1_bpr.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_r1_fetch -o bpr -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_r1_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_r1_write -o bpr -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_r1_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_r1_l2 -o bpr -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_r1_l2.log 2>&1
cd $R
find gpurun_out -name "*.csv" | head -40
tail -3 gpurun_out/smoke_r1.log; tail -12 gpurun_out/pytest_r1.log; cat gpurun_out/bench_r1.json
