"""Round-6 study (a'), after scripts/r6_lr005_width.py: |Qb| of the reference path is 183.0 at 8 / 64 / 128 / 256 workers and 147 for the walk under every
im_blocks x im_max_stale -- neither the width nor the burst order.  This dumps the bias vectors themselves (refbench: lr 0.05 -> 0.0001, 10 epochs) so that
they can be compared item by item: WHICH rows carry the difference.  env WHO = oracle (CPU, 8 workers) | hip (modes from MODES, JSON) ; EPOCHS."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import test_bpr_gate_gpu as G  # noqa: E402

who = os.environ.get("WHO", "oracle")
epochs = int(os.environ.get("EPOCHS", "10"))
total = int(os.environ.get("TOTAL", "10"))           # num_iters of the schedule (the lr decays over these)
csr = G._csr()
opt = bench.bpr_options(total, lr=0.05, min_lr=0.0001)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
if who == "oracle":
    from oracle import oracle as orc
    ((o, P, Q, Qb),) = G._run_oracles(orc, csr, opt, (int(os.environ.get("WORKERS", "8")),), epochs)
    tag = "oracle"
else:
    modes = json.loads(os.environ.get("MODES", "{}"))
    obj, P, Q, Qb = G._run_hip(csr, opt, epochs, modes)
    tag = "hip" + os.environ.get("TAG", "")
cnt = np.bincount(csr.keys, minlength=csr.num_items)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "r6_qb_%s_e%d.npz" % (tag, epochs)), Qb=Qb.ravel(), qn=np.linalg.norm(Q, axis=1), cnt=cnt,
                    pn=np.linalg.norm(P, axis=1))
print(tag, epochs, "Qb", float(np.linalg.norm(Qb)), "Q", float(np.linalg.norm(Q)), "P", float(np.linalg.norm(P)))
