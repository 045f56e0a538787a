import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from conftest import bpr_opt
from buffalo_amd import synth
from buffalo_amd.backend import CyBPR
csr, vali = synth.planted(600, 400, d_true=6, density=0.06, seed=7)
d, vdim = 16, 32
opt = bpr_opt(d=d, lr=0.05, min_lr=0.01, num_iters=30, random_seed=7, num_workers=4, reg_u=0.01, reg_i=0.01, reg_j=0.01, reg_b=0.01)
P0, Q0, Qb0 = synth.init_factors(600, 400, d, seed=7)
print("nnz", csr.nnz, "base", H.ndcg_at_k(P0, Q0, csr, vali, Qb=Qb0))
for name, modes in [("atomic", dict(hogwild_atomic=1, chunk=64)), ("write-through", dict(hogwild_atomic=0, chunk=64)),
                    ("write-through nopf", dict(hogwild_atomic=0, chunk=64, prefetch=0)),
                    ("hybrid hot=64", dict(hogwild_atomic=2, chunk=64, hot_items=64)),
                    ("write-through chunk1024", dict(hogwild_atomic=0, chunk=1024)),
                    ("write-through chunk4096", dict(hogwild_atomic=0, chunk=4096)),
                    ("sequential", dict(sequential=1))]:
    P, Q, Qb = H.pad(P0, vdim), H.pad(Q0, vdim), Qb0.copy()
    H.run_hip_sgd(CyBPR, opt, csr, P, Q, Qb, epochs=30, modes=modes, resident=True)
    print("%-18s ndcg %.4f  |P| %.3f |Q| %.3f |Qb| %.3f" % (name, H.ndcg_at_k(P[:, :d], Q[:, :d], csr, vali, Qb=Qb),
          np.linalg.norm(P), np.linalg.norm(Q), np.linalg.norm(Qb)), flush=True)
