#!/bin/bash
# round 3, GPU call 14: the bench line as the driver runs it (default flags), then the rocprofv3 passes of the same command
# (kernel stats of a run that prints the line; separate --pmc passes) -> profiles/r03_bench_*, profiles/pmc_latest.json.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c14; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; echo "bench rc=$?"; tail -3 $O/bench_default.time
python - <<'P'
import json
d=json.loads(open("gpurun_out/r3c14/bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","steps")}, {k:d["roofline"][k] for k in ("frac","kernel_ms","achieved")}, d["roofline"]["measured_stream"].get("frac_of_triad"))
print("cpu", d["cpu_baseline"]["value"])
for k,v in d["extra"].items():
    print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,(list,dict,str))}, ("ERROR "+v["error"]) if "error" in v else "")
P
bash scripts/gpu_profile.sh 2>&1 | tail -40
