"""Round-6 study (a) of the lr-0.05 question: is the walk's offset at the reference benchmark's BPRMF setting
(/root/reference/benchmark/models.py:86-93: lr 0.05 -> 0.0001, 10 iterations) the WIDTH of the schedule or the ORDER of the item-major walk?

Part 1 (CPU, on the GPU box's host cores): the oracle's threaded Hogwild (the reference path, bpr.cc:72-188) at WORKERS = 8, 64, 128, 256
threads from the same initial factors -- does |Qb| of the reference path itself move from 183 towards the walk's 148 as the pool widens?
Part 2 (GPU): the walk under im_blocks x im_max_stale, one run each, against every oracle of part 1.

env: CASE (refbench | lr0.05), WORKERS ("8,64,128,256"), SWEEP (JSON list of knob dicts; default the blocks x stale grid), EPOCHS.
Output: one JSON line per run + gpurun_out/r6_lr005_width_<case>.json."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import test_bpr_gate_gpu as G  # noqa: E402
from oracle import oracle as orc  # noqa: E402

case = os.environ.get("CASE", "refbench")
kw, default_epochs = {"lr0.05": (dict(lr=0.05, min_lr=0.05), 24), "refbench": (dict(lr=0.05, min_lr=0.0001), 10),
                      "bench": (dict(lr=0.002, min_lr=0.0001), 3)}[case]
epochs = int(os.environ.get("EPOCHS", str(default_epochs)))
workers = [int(w) for w in os.environ.get("WORKERS", "8,64,128,256").split(",") if w]
csr = G._csr()
users = np.random.default_rng(1).choice(csr.num_users, 2000, replace=False)
eu, ep, en = G._eval_set(csr)
opt = bench.bpr_options(epochs, **kw)
out = {"case": case, "epochs": epochs, "oracles": [], "hip": []}
tops = {}
# ---- part 1: the reference path at several pool widths (two pools side by side at a time) ----
for i in range(0, len(workers), 2):
    pair = workers[i:i + 2]
    t0 = time.time()
    objs = G._run_oracles(orc, csr, opt, pair, epochs)
    dt = time.time() - t0
    for w, (o, P, Q, Qb) in zip(pair, objs):
        top = G._top10(P, Q, Qb, users)
        m = G._metrics(lambda: o.compute_loss(eu, ep, en), P, Q, Qb)
        m["prec10"] = G._precision10(csr, top, users)
        m["workers"] = w
        m["pair_seconds"] = round(dt, 1)
        tops[w] = top
        out["oracles"].append(m)
        print("oracle", json.dumps(m), flush=True)
    del objs
ws = sorted(tops)
for a in ws:
    for b in ws:
        if a < b:
            print("oracle overlap %d~%d %.3f" % (a, b, G._overlap(tops[a], tops[b])), flush=True)
# ---- part 2: the walk ----
if os.environ.get("SKIP_HIP", "0") != "1":
    if "SWEEP" in os.environ:
        sweep = json.loads(os.environ["SWEEP"])
    else:
        sweep = [{}] + [{"im_blocks": b, "im_max_stale": s} for b in (8, 16, 32, 64) for s in (4, 8, 16)]
    for modes in sweep:
        obj, P, Q, Qb = G._run_hip(csr, opt, epochs, modes)
        top = G._top10(P, Q, Qb, users)
        m = G._metrics(lambda: obj.compute_loss(eu, ep, en), P, Q, Qb)
        m["prec10"] = G._precision10(csr, top, users)
        m["overlap"] = {str(w): round(G._overlap(top, tops[w]), 3) for w in ws}
        st = obj.stats()
        m["kernel_ms_per_launch"] = st["kernel_ms"] / max(1, st["launches"])
        m["modes"] = modes
        out["hip"].append(m)
        print("hip", json.dumps(m), flush=True)
        del obj
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r6_lr005_width_%s.json" % case.replace(".", "")), "w"), indent=1)
