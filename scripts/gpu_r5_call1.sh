#!/bin/bash
# GPU call 1 (round 5): the drift-delta walk (conflict-free parity, gate study at lr 0.05 against the stored oracle pair), the ALS heavy + deferred
# rows case, the new bench line.
O=gpurun_out/r5c1; mkdir -p $O
timeout 600 python -m pytest tests/test_bpr_gpu.py -x -q -k "conflict_free or single_wave or statistical" > $O/pytest_bpr.txt 2>&1; echo "pytest bpr rc=$?"; tail -3 $O/pytest_bpr.txt
timeout 300 python -m pytest tests/test_als_gpu.py -x -q -k "heavy" > $O/pytest_als.txt 2>&1; echo "pytest als rc=$?"; tail -3 $O/pytest_als.txt
export REPS=2
export SETTINGS='[{}, {"im_drift_delta":0}, {"im_user_lr_max":1000,"xcd_stiff_p":50}]'
CASE=lr0.05 timeout 600 python scripts/gate_knob_study.py > $O/study_lr005.txt 2>&1; echo "study rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r5c1/study_lr005.txt",):
    for line in open(f):
        if line.startswith("oracle"):
            print(line.strip()[:300])
        elif line.startswith("{"):
            m = json.loads(line)
            print("%-70s loss %.4f P %.2f Q %.3f Qb %.2f p10 %.3f ov %.3f k %.3f aux %.2f" % (json.dumps(m["modes"]), m["loss"], m["P"], m["Q"], m["Qb"], m["prec10"], m["overlap"], m["kernel_ms_per_launch"], m["aux_ms_per_epoch"]))
        elif "Error" in line or "error" in line:
            print(line.strip()[:300])
PY
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"
tail -c 4200 $O/bench.out | tail -1
tail -5 $O/bench.err
