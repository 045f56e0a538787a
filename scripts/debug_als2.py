import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from conftest import als_opt
from oracle import oracle as orc
from buffalo_amd import synth
from buffalo_amd.backend import CyALS
csr = synth.generate(*synth.SHAPES["ml100k"], seed=7, vals="counts")
t = csr.transpose()
for d, optimizer in ((32, "llt"), (32, "manual_cg"), (128, "ialspp")):
    opt = als_opt(d=d, optimizer=optimizer, compute_loss_on_training=True)
    rng = np.random.default_rng(7)
    P = np.abs(rng.normal(scale=1.0 / d, size=(csr.num_users, d))).astype(np.float32)
    Q = np.abs(rng.normal(scale=1.0 / d, size=(csr.num_items, d))).astype(np.float32)
    Po, Qo = P.copy(), Q.copy()
    o = orc.OracleALS(); assert o.init(H.write_opt(opt)); o.initialize_model(Po, Qo)
    g = CyALS(); assert g.init(H.write_opt(dict(opt, accelerator=True))); g.initialize_model(P, Q)
    g.set_placeholder(csr.indptr, t.indptr, csr.nnz + 1)
    for axis, mat in ((0, csr), (1, t)):
        o.precompute(axis); g.precompute(axis)
        lo = o.partial_update(0, mat.num_users, mat.indptr, mat.keys, mat.vals, axis)
        lg = g.partial_update(0, mat.num_users, mat.indptr, mat.keys, mat.vals, axis)
        X, Xo = (P, Po) if axis == 0 else (Q, Qo)
        err = np.abs(X - Xo).max(1) / np.abs(Xo).max()
        deg = np.diff(np.concatenate([[0], mat.indptr]))
        worst = np.argsort(-err)[:5]
        print(d, optimizer, "axis", axis, "relerr %.3e" % H.relerr(X, Xo), "loss", lo, lg, "worst rows", worst, "deg", deg[worst], "err", err[worst],
              "| err by deg<=64: %.2e  deg>64: %.2e" % (err[deg <= 64].max(), err[deg > 64].max() if (deg > 64).any() else 0), flush=True)
        X[:] = Xo
        g.initialize_model(P, Q); g.set_placeholder(csr.indptr, t.indptr, csr.nnz + 1)
