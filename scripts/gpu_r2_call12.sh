#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c12; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_topk_gpu.py -m gpu -q -s --maxfail=60 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_topk -o topk -- python $GRAFT_REPO_ROOT/scripts/bench_extra.py topk > $GRAFT_REPO_ROOT/$O/topk_prof.log 2>&1)
find $O/prof_topk -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/topk_kernel_stats.csv
rm -rf $O/prof_topk
timeout 400 python bench.py --steps 50 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
grep -E "passed|failed|FAILED|rc=" $O/pytest.log | tail -10
grep -E "^topk" $O/topk_prof.log | cut -c1-420
head -16 $O/topk_kernel_stats.csv | cut -c1-160
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print("bench", d["value"]/1e9, d["ms_per_step"], d["roofline"]["frac"])
for k,v in d["extra"].items():
    print(k, json.dumps(v)[:900])
PY
tail -3 $O/bench.err
