#!/bin/bash
# GPU call 2 (round 5): drift rows with the bias chip-wide and the factor row on (chip-wide row + delta) or in the replica -- conflict-free parity,
# gate studies at constant lr 0.05 and on the reference benchmark's schedule (lr 0.05 -> 0.0001, 10 epochs); bench.py --gpus 2 self-launched on one GPU.
O=gpurun_out/r5c2; mkdir -p $O
timeout 600 python -m pytest tests/test_bpr_gpu.py -x -q -k "conflict_free" > $O/pytest_bpr.txt 2>&1; echo "pytest bpr rc=$?"; tail -3 $O/pytest_bpr.txt
export REPS=2
export SETTINGS='[{"im_drift_delta":1}, {"im_drift_delta":2}, {"im_drift_delta":0}]'
CASE=lr0.05 timeout 600 python scripts/gate_knob_study.py > $O/study_lr005.txt 2>&1; echo "study lr0.05 rc=$?"
CASE=refbench timeout 600 python scripts/gate_knob_study.py > $O/study_refbench.txt 2>&1; echo "study refbench rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r5c2/study_lr005.txt", "gpurun_out/r5c2/study_refbench.txt"):
    print(f)
    for line in open(f):
        if line.startswith("oracle"):
            print(line.strip()[:300])
        elif line.startswith("{"):
            m = json.loads(line)
            print("%-70s loss %.4f P %.2f Q %.3f Qb %.2f p10 %.3f ov %.3f k %.3f aux %.2f" % (json.dumps(m["modes"]), m["loss"], m["P"], m["Q"], m["Qb"], m["prec10"], m["overlap"], m["kernel_ms_per_launch"], m["aux_ms_per_epoch"]))
        elif "Error" in line or "error" in line:
            print(line.strip()[:300])
PY
export BFH_DEVICE_OVERRIDE=0 BFH_COMM_TRANSPORT=shm
for w in bpr als; do
  timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 --workload $w > $O/n2_$w.out 2> $O/n2_$w.err; echo "n2 $w rc=$?"; tail -1 $O/n2_$w.out | head -c 1500; echo; tail -3 $O/n2_$w.err
done
timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --workload warp_c5 --c5-users 400000 > $O/n2_warp.out 2> $O/n2_warp.err; echo "n2 warp rc=$?"; tail -1 $O/n2_warp.out | head -c 1500; echo; tail -3 $O/n2_warp.err
