#!/bin/bash
# round 4, GPU call 22 (final code): the whole -m gpu suite as the driver runs it, smoke, the default bench line, rocprofv3 passes of the same command
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c22; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=6 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|FAILED|rc=" $O/pytest.log | tail -5
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; echo "bench rc=$?"; tail -3 $O/bench_default.time
python - <<'P'
import json
d=json.loads(open("gpurun_out/r4c22/bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","steps")})
print({k:v for k,v in d["roofline"].items() if not isinstance(v,(dict,list,str))})
print({k:v for k,v in d["cpu_baseline"].items() if not isinstance(v,(dict,list,str))})
for k,v in d["extra"].items():
    if "error" in v: print(k, "ERROR", v["error"])
P
bash scripts/gpu_profile.sh 2>&1 | grep -E "bpr_item_major|als_pc|hbm_bytes|\"value\"" | cut -c1-300 | tail -8
