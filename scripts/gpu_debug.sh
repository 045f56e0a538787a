export TMPDIR=/tmp
mkdir -p gpurun_out
for dbg in 0 1 2 3 5 7; do echo "ALS_DEBUG=$dbg"; ALS_DEBUG=$dbg timeout 300 python scripts/bench_extra.py als 2>&1 | grep "^als" | cut -c1-120; done > gpurun_out/als_ablate.log 2>&1
cat gpurun_out/als_ablate.log
