#!/bin/bash
# round 3, GPU call 5: what changed since call 4 (bias steps in double, exchange timers, dist harness split, Par* stand-ins moved,
# smoke envelope), configs[4] shape with the popularity law under the profiler, and bench.py --gpus 2 / 8 on ONE GPU through the
# library's exchange (shared-memory test transport) to see the N > 1 line and its breakdown before an 8-GPU node does.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c5; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bpr_gpu.py tests/test_topk_gpu.py tests/test_comm_gpu.py tests/test_comm_ranks_gpu.py tests/test_warp_gpu.py tests/test_front_gpu.py "tests/test_als_gpu.py::test_half_epochs_match_oracle" tests/test_als_gpu.py -k "not config3 and not full_size" -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
for N in 2 8; do
  BFH_DEVICE_OVERRIDE=0 BFH_DIST_BACKEND=gloo BFH_COMM_TRANSPORT=shm timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $N --steps 10 --warmup 3 > $O/bench_n${N}_shm_one_gpu.json 2> $O/bench_n${N}.err
  echo "bench N=$N rc=$?"; tail -c 1500 $O/bench_n${N}_shm_one_gpu.json; tail -3 $O/bench_n${N}.err
done
# configs[4] shape with popularity: one profiler pass set (reuses scripts/gpu_profile_warp.sh's layout for c5 only)
R=$GRAFT_REPO_ROOT; W=$R/gpurun_out/warp_prof_c5skew; rm -rf $W; mkdir -p $W; cd /tmp
CMD="python $R/scripts/run_warp.py shape=c5 epochs=6"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $W/c5/stats -o p -- $CMD out=$W/c5_epochs.json > $W/c5_stats.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $W/c5/pmc_fetch -o p -- $CMD > $W/c5_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $W/c5/pmc_write -o p -- $CMD > $W/c5_write.log 2>&1
python $R/scripts/pmc_kernels.py $W/c5 6 $W/c5.json > $W/c5_summary.txt 2>&1
for f in $(find $W/c5/stats -name "*kernel_stats.csv"); do cp $f $W/c5_kernel_stats.csv; done
find $W/c5 -name "*.csv" -size +2M -delete
grep run_warp $W/c5_stats.log | cut -c1-420; cat $W/c5_summary.txt
