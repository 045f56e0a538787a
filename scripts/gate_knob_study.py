"""GPU side of the gate study (env CASE = "lr0.05" | "bench", SETTINGS = JSON list of knob dicts, EPOCHS): the item-major walk under several knob settings (and the all-atomic user-major walk),
three runs each, against the stored oracle reference (scripts/gate_oracle_ref.py).  Prints one line per run."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import test_bpr_gate_gpu as G  # noqa: E402

case = os.environ.get("CASE", "lr0.05")
kw, default_epochs = {"lr0.05": (dict(lr=0.05, min_lr=0.05), 24), "bench": (dict(lr=0.002, min_lr=0.0001), 3),
                      "refbench": (dict(lr=0.05, min_lr=0.0001), 10)}[case]
ref = np.load(os.path.join(ROOT, "scripts", "data", "gate_%s_oracle.npz" % case.replace(".", "")))
users = ref["users"]
csr = G._csr()
eu, ep, en = G._eval_set(csr)
epochs = int(os.environ.get("EPOCHS", str(default_epochs)))
opt = bench.bpr_options(epochs, **kw)
print("oracle a", ref["metrics_a"], "b", ref["metrics_b"], "a~b overlap %.3f" % G._overlap(ref["top_a"], ref["top_b"]), flush=True)
SETTINGS = json.loads(os.environ["SETTINGS"]) if "SETTINGS" in os.environ else [{}, {"im_max_stale": 16}, {"im_max_stale": 4}, {"xcd_sync_updates": 1 << 21}, {"xcd_sync_updates": 1 << 20, "im_max_stale": 16},
            {"im_blocks": 16}, {"im_blocks": 2}, {"xcd_hot_tau": 30}, {"im_drift_budget": 250}, {"hogwild_atomic": 1}]
rows = []
for modes in SETTINGS:
    for rep in range(int(os.environ.get("REPS", "3"))):
        obj, P, Q, Qb = G._run_hip(csr, opt, epochs, modes)
        top = G._top10(P, Q, Qb, users)
        m = G._metrics(lambda: obj.compute_loss(eu, ep, en), P, Q, Qb)
        m["prec10"] = G._precision10(csr, top, users)
        m["overlap"] = 0.5 * (G._overlap(top, ref["top_a"]) + G._overlap(top, ref["top_b"]))
        st = obj.stats()
        m["kernel_ms_per_launch"] = st["kernel_ms"] / max(1, st["launches"])
        m["aux_ms_per_epoch"] = st["aux_ms"] / epochs
        m["modes"] = modes
        rows.append(m)
        print(json.dumps(m), flush=True)
        del obj
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "gate_knob_study_%s.json" % case), "w"), indent=1)
