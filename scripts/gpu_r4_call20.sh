#!/bin/bash
# GPU call 20: merge weights, second pass -- (a) lr 0.05 with the drift rule relaxed (replicas in use again), (b) which weight matters at the bench lr.
mkdir -p gpurun_out/r4c20
export REPS=2
export SETTINGS='[{"im_drift_budget":0}, {"im_drift_budget":0,"xcd_stiff_q":250,"xcd_stiff_b":250}, {"im_drift_budget":0,"xcd_stiff_q":1000,"xcd_stiff_b":1000},
 {"im_drift_budget":4000,"xcd_stiff_q":250,"xcd_stiff_b":250}, {"im_drift_budget":4000,"xcd_stiff_q":1000,"xcd_stiff_b":1000}, {"im_drift_budget":4000},
 {"im_drift_budget":16000,"xcd_stiff_q":250,"xcd_stiff_b":250}, {"im_drift_budget":16000,"xcd_stiff_q":1000,"xcd_stiff_b":1000},
 {"im_drift_budget":0,"xcd_stiff_q":250,"xcd_stiff_b":250,"xcd_sync_updates":2097152}, {"im_drift_budget":0,"xcd_stiff_q":1000,"xcd_stiff_b":1000,"xcd_sync_updates":2097152},
 {"im_blocks":1}, {"im_blocks":2}, {"im_blocks":2,"im_drift_budget":4000,"xcd_stiff_q":250,"xcd_stiff_b":250}]'
CASE=lr0.05 timeout 900 python scripts/gate_knob_study.py > gpurun_out/r4c20/study_lr005.txt 2>&1
echo "lr0.05 rc=$?"
export SETTINGS='[{}, {"xcd_stiff_b":250}, {"xcd_stiff_q":250}, {"xcd_stiff_p":250}, {"xcd_stiff_b":500}, {"xcd_stiff_q":500,"xcd_stiff_b":500,"xcd_stiff_p":500}, {"xcd_stiff_b":250,"im_user_hybrid":0}, {"im_user_hybrid":0}]'
CASE=bench timeout 300 python scripts/gate_knob_study.py > gpurun_out/r4c20/study_bench.txt 2>&1
echo "bench rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r4c20/study_lr005.txt", "gpurun_out/r4c20/study_bench.txt"):
    print(f)
    for line in open(f):
        if line.startswith("oracle"):
            print(line.strip()[:400])
        elif line.startswith("{"):
            m = json.loads(line)
            print("%-100s loss %.4f P %.2f Q %.3f Qb %.2f p10 %.3f ov %.3f k %.3f aux %.2f" % (json.dumps(m["modes"]), m["loss"], m["P"], m["Q"], m["Qb"], m["prec10"], m["overlap"], m["kernel_ms_per_launch"], m["aux_ms_per_epoch"]))
        elif "Error" in line or "error" in line:
            print(line.strip()[:300])
PY
