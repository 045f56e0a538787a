#!/bin/bash
# round 3, GPU call 19 (final state): the whole -m gpu suite as the driver runs it, smoke, the default bench line, and the rocprofv3
# passes of the same command (kernel stats + PMC) for profiles/.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c19; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=6 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|FAILED|rc=" $O/pytest.log | tail -5
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; echo "bench rc=$?"; tail -3 $O/bench_default.time
python - <<'P'
import json
d=json.loads(open("gpurun_out/r3c19/bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","steps")}, {k:d["roofline"][k] for k in ("frac","kernel_ms","achieved")}, d["roofline"]["measured_stream"].get("frac_of_triad"))
for k,v in d["extra"].items():
    print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,(list,dict,str))}, ("ERROR "+v["error"]) if "error" in v else "")
P
bash scripts/gpu_profile.sh 2>&1 | grep -E "bpr_item_major|hbm_bytes|\"value\"" | cut -c1-400 | tail -8
