#!/bin/bash
# round 2, GPU call 2: full GPU suite with the new paths, ALS wall-time probe, bench line, secondary timings
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c2; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -s --maxfail=10 --deselect tests/test_bpr_gate_gpu.py -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python scripts/als_overhead_probe.py > $O/als_probe_plain.log 2>&1
timeout 300 python scripts/als_overhead_probe.py torch > $O/als_probe_torch.log 2>&1
timeout 600 python bench.py --steps 100 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
timeout 400 python scripts/bench_extra.py bpr_adagrad bpr_pcie > $O/extra.log 2>&1
timeout 400 python -m pytest "tests/test_bpr_gate_gpu.py::test_item_major_tracks_threaded_oracle_at_baseline_scale[lr0.05]" -m gpu -q -s -p no:cacheprovider > $O/gate_lr05.log 2>&1
grep -E "passed|failed|FAILED|rc=" $O/pytest.log | tail -15; tail -4 $O/als_probe_plain.log; tail -4 $O/als_probe_torch.log; head -c 1500 $O/bench.json; echo; python - <<'PY'
import json
d=json.load(open("gpurun_out/r2c2/bench.json"))
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ("epoch_ms","kernel_ms_per_epoch","mfma","epochs","error")}) for k,v in d.get("extra",{}).items()})
PY
tail -3 $O/extra.log; grep -E "^\[|oracle-|hip  |top-10|passed|failed" $O/gate_lr05.log | tail -8
