#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c23; mkdir -p $O
export TMPDIR=/tmp
BFH_DEVICE_OVERRIDE=0 BFH_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_2ranks_gloo.log 2>&1
echo "2rank rc=$?" >> $O/bench_2ranks_gloo.log
BFH_DEVICE_OVERRIDE=0 BFH_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 8 --steps 3 --warmup 1 > $O/bench_8ranks_gloo.log 2>&1
echo "8rank rc=$?" >> $O/bench_8ranks_gloo.log
tail -3 $O/bench_2ranks_gloo.log | cut -c1-700; tail -3 $O/bench_8ranks_gloo.log | cut -c1-700
