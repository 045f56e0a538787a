#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c4; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -s --maxfail=10 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
BFH_DEVICE_OVERRIDE=0 BFH_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_2ranks_gloo.log 2>&1
echo "2rank rc=$?" >> $O/bench_2ranks_gloo.log
timeout 300 python scripts/shard_times.py > $O/shards.log 2>&1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
timeout 500 python scripts/bench_extra.py warp_c5 > $O/warp_c5.log 2>&1
grep -E "passed|failed|FAILED|rc=" $O/pytest.log | tail -12; cat $O/smoke.log | tail -2; tail -3 $O/bench_2ranks_gloo.log | cut -c1-600; tail -4 $O/shards.log; head -c 700 $O/bench.json; echo; tail -4 $O/warp_c5.log | cut -c1-500
