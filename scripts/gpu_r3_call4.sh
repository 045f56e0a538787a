#!/bin/bash
# round 3, GPU call 4: (1) the quarantined whole-run tests (first device measurement of the per-call envelope and of the well-posed
# case), (2) the per-row front trace for profiles/, (3) the WARP roofline passes, (4) a short bench line with the new extras.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c4; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_trained_models_ref.py -m gpu_unmeasured -q -s -p no:cacheprovider > $O/pytest_trained.log 2>&1
echo "pytest rc=$?" >> $O/pytest_trained.log
grep -E "hip~golden|call [0-9]|passed|failed|rc=" $O/pytest_trained.log | tail -40
timeout 300 python scripts/als_cg_diag.py front > $O/als_front_trace.txt 2>&1
grep "rows by" $O/als_front_trace.txt | head -8
bash scripts/gpu_profile_warp.sh 2>&1 | tail -60
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open("gpurun_out/r3c4/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"])
for k,v in d["extra"].items():
    print(k, {a:b for a,b in v.items() if not isinstance(b,(list,dict))})
    if "epochs" in v and isinstance(v["epochs"], list):
        for e in v["epochs"]:
            print("    ", {a:(round(b,4) if isinstance(b,float) else b) for a,b in e.items() if a!="implemented_model_bytes"})
P
