#!/bin/bash
# Round 6, final measurement pass on the round's code: the whole GPU suite, smoke, the rocprofv3 passes of the bench command (kernel stats of the run that prints
# the line; FETCH / WRITE / TCC / MFMA counters in separate passes), then the default bench line (250 steps, CPU baselines) -> gpurun_out/r6final
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6final; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu --durations=8 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -14 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt
PROF_DIR=r6prof bash scripts/gpu_profile.sh > $O/profile_stdout.txt 2>&1; echo "profile rc=$?"; tail -30 $O/profile_stdout.txt | cut -c1-220
cp gpurun_out/r6prof/pmc_latest.json $O/ 2>/dev/null
cp gpurun_out/r6prof/pmc_latest.json profiles/pmc_latest.json 2>/dev/null   # (on the box: the default bench below stamps its traffic from THIS library's counter passes)
timeout 900 python bench.py > $O/bench_default.out 2> $O/bench_default.err; echo "bench rc=$?"; tail -1 $O/bench_default.out | wc -c; tail -1 $O/bench_default.out
cp bench_extra.json $O/bench_extra_default.json 2>/dev/null
cp gpurun_out/r6prof/bench_under_rocprof.json $O/ 2>/dev/null
for f in $(find gpurun_out/r6prof/stats -name "*kernel_stats.csv"); do cp $f $O/bench_rocprofv3_kernel_stats.csv; done
# the driver's N > 1 launch (python -m torch.distributed.run ...) rehearsed on the one GPU over the test transport: the line must carry rccl_ranks / transport
for N in 2 8; do
  BFH_DEVICE_OVERRIDE=0 BFH_COMM_TRANSPORT=shm timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $N --steps 10 --warmup 3 > $O/bench_n${N}_shm_one_gpu.json 2> $O/bench_n${N}.err
  echo "bench N=$N rc=$?"; python - <<P
import json
d=json.loads(open("$O/bench_n${N}_shm_one_gpu.json").read().strip().splitlines()[-1])
print(d["n_gpus"], d.get("rccl_ranks"), d.get("transport"), d["value"], d["ms_per_step"], d["config"]["parallelism"][:100], d.get("breakdown"))
P
done
