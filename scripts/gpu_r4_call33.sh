#!/bin/bash
# GPU call 33: the library as it stands at the end of the round: smoke + the bench line without extras
mkdir -p gpurun_out/r4c33
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4c33/smoke.log 2>&1; echo "smoke rc=$?"
timeout 100 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra > gpurun_out/r4c33/bench.json 2> gpurun_out/r4c33/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r4c33/bench.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','steps')}, d['roofline']['kernel_ms'], d['roofline']['frac'])"
