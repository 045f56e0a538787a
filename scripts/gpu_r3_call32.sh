#!/bin/bash
# round 3, GPU call 32: the committed state as the driver runs it (-x) with the split-f16 ALS pass, smoke, and the default bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c32; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=8 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|FAILED|rc=" $O/pytest.log | tail -4
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<P
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
a=[e for e in d.get("extra",{}).values() if isinstance(e,dict) and "ALS" in str(e.get("config",""))]
for e in a: print(e.get("epoch_ms"), e.get("kernel_ms_per_epoch"), e.get("mfma"))
P
