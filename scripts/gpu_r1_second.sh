set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
timeout 120 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke_r1.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_r1.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_r1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_r1.log
# BPR knob sweep (one JSON line per variant)
: > gpurun_out/sweep_r1.jsonl
for m in "" "--mode hogwild_atomic=0" "--mode prefetch=0" "--mode chunk=64" "--mode chunk=128" "--mode chunk=512" "--mode chunk=1024" "--mode waves_per_cu=16" "--mode waves_per_cu=24" "--mode waves_per_cu=48" "--mode waves_per_cu=64" "--mode hogwild_atomic=0 --mode prefetch=0"; do
  echo "## $m" >> gpurun_out/sweep_r1.jsonl
  timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $m >> gpurun_out/sweep_r1.jsonl 2>> gpurun_out/sweep_r1.err
done
# rocprofv3 kernel trace of the default bench
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1_bpr -o bpr -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_r1_bpr.log 2>&1
# PMC passes (separate runs; no trace domains besides kernel-trace)
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_r1_fetch -o bpr -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_r1_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_r1_write -o bpr -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_r1_write.log 2>&1
cd $R
find gpurun_out -name "*.csv" | head -30
tail -3 gpurun_out/smoke_r1.log; tail -30 gpurun_out/pytest_r1.log; cat gpurun_out/sweep_r1.jsonl | cut -c1-400
