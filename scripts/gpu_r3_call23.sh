#!/bin/bash
# round 3, GPU call 23: the committed state once more as the driver runs it (-x), smoke, and bench.py --gpus 2 / 8 through the
# library's exchange on one GPU (shm transport) with the final kernels.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c23; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=5 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|FAILED|rc=" $O/pytest.log | tail -4
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
for N in 2 8; do
  BFH_DEVICE_OVERRIDE=0 BFH_DIST_BACKEND=gloo BFH_COMM_TRANSPORT=shm timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $N --steps 10 --warmup 3 > $O/bench_n${N}_shm_one_gpu.json 2> $O/bench_n${N}.err
  echo "bench N=$N rc=$?"; python - <<P
import json
d=json.loads(open("$O/bench_n${N}_shm_one_gpu.json").read().strip().splitlines()[-1])
print(d["n_gpus"], d["config"]["parallelism"][:90], d["breakdown"])
P
done
