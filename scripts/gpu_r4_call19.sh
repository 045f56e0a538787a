#!/bin/bash
# GPU call 19: intra-GPU merge weights (VERDICT r03 #6a/b): lr 0.05 / 24 epochs against the stored oracle pair, then the bench case.
mkdir -p gpurun_out/r4c19
export REPS=2
export SETTINGS='[{}, {"xcd_stiff_b":50}, {"xcd_stiff_b":100}, {"xcd_stiff_b":250}, {"xcd_stiff_b":1000},
 {"xcd_stiff_q":50,"xcd_stiff_b":100}, {"xcd_stiff_q":100,"xcd_stiff_b":100}, {"xcd_stiff_q":250,"xcd_stiff_b":250}, {"xcd_stiff_q":1000,"xcd_stiff_b":1000},
 {"im_user_lr_max":1000}, {"im_user_lr_max":1000,"xcd_stiff_p":50}, {"im_user_lr_max":1000,"xcd_stiff_p":100}, {"im_user_lr_max":1000,"xcd_stiff_p":250}, {"im_user_lr_max":1000,"xcd_stiff_p":1000},
 {"im_user_lr_max":1000,"xcd_stiff_p":100,"xcd_stiff_q":100,"xcd_stiff_b":100}, {"im_user_lr_max":1000,"xcd_stiff_p":250,"xcd_stiff_q":250,"xcd_stiff_b":250}]'
CASE=lr0.05 timeout 900 python scripts/gate_knob_study.py > gpurun_out/r4c19/study_lr005.txt 2>&1
echo "lr0.05 rc=$?"
export SETTINGS='[{}, {"xcd_stiff_q":250,"xcd_stiff_b":250,"xcd_stiff_p":250}, {"xcd_stiff_q":1000,"xcd_stiff_b":1000,"xcd_stiff_p":1000}]'
CASE=bench timeout 300 python scripts/gate_knob_study.py > gpurun_out/r4c19/study_bench.txt 2>&1
echo "bench rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r4c19/study_lr005.txt", "gpurun_out/r4c19/study_bench.txt"):
    print(f)
    for line in open(f):
        if line.startswith("oracle"):
            print(line.strip()[:400])
        elif line.startswith("{"):
            m = json.loads(line)
            print("%-90s loss %.4f P %.1f Q %.2f Qb %.2f p10 %.3f ov %.3f k %.3f aux %.2f" % (json.dumps(m["modes"]), m["loss"], m["P"], m["Q"], m["Qb"], m["prec10"], m["overlap"], m["kernel_ms_per_launch"], m["aux_ms_per_epoch"]))
        elif "Error" in line or "error" in line:
            print(line.strip()[:300])
PY
