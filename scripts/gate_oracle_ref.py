"""CPU side of the gate study (argv[1]: "lr0.05" = 24 epochs at constant lr 0.05, "bench" = bench.py's options, 3 epochs): runs the
oracle's threaded Hogwild (8 and 16 workers) on
BASELINE configs[1] and stores what the GPU side compares against (norms, sampled loss, the top-10 lists of 2,000 users).
The oracle needs no GPU, so it runs wherever there are cores; scripts/gate_knob_study.py then runs on the GPU box."""
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

case = sys.argv[1] if len(sys.argv) > 1 else "lr0.05"
kw, epochs, workers = {"lr0.05": (dict(lr=0.05, min_lr=0.05), 24, (8, 16)), "bench": (dict(lr=0.002, min_lr=0.0001), 3, (8, 16)),
                       # the reference's own BPRMF benchmark setting (benchmark/models.py:86-93): lr 0.05 decaying to min_lr 0.0001 over 10 iterations
                       "refbench": (dict(lr=0.05, min_lr=0.0001), 10, (8, 16))}[case]
csr = G._csr()
users = np.random.default_rng(1).choice(csr.num_users, 2000, replace=False)
eu, ep, en = G._eval_set(csr)
opt = bench.bpr_options(epochs, **kw)
t0 = time.time()
objs = G._run_oracles(orc, csr, opt, workers, epochs)
out = {"users": users}
for name, (o, P, Q, Qb) in zip(("a", "b"), objs):
    top = G._top10(P, Q, Qb, users)
    m = G._metrics(lambda: o.compute_loss(eu, ep, en), P, Q, Qb)
    m["prec10"] = G._precision10(csr, top, users)
    print(name, m, flush=True)
    out["top_" + name] = top
    out["metrics_" + name] = np.array([m["loss"], m["P"], m["Q"], m["Qb"], m["prec10"]])
print("oracle a~b overlap %.3f, %.0f s" % (G._overlap(out["top_a"], out["top_b"]), time.time() - t0))
np.savez_compressed(os.path.join(ROOT, "scripts", "data", "gate_%s_oracle.npz" % case.replace(".", "")), **out)
