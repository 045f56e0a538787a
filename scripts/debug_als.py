"""Debug helper: oracle vs HIP vs float64 ground truth for one ALS half-epoch."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from conftest import als_opt, tiny_csr
from oracle import oracle as orc
from buffalo_amd.backend import CyALS


def ialspp_f64(P, Q, FF, u, keys, vals, alpha, reg, block_size, tol=1e-10):
    D = Q.shape[1]
    p = P[u].astype(np.float64).copy()
    Qd = Q.astype(np.float64); FFd = FF.astype(np.float64)
    Y = np.array([p @ Qd[c] for c in keys])
    bs0 = min(D, block_size)
    for bb in range(0, D, bs0):
        bs = bs0 if bb + bs0 < D else D - bb
        gram = FFd[:, bb:bb + bs]
        A = gram[bb:bb + bs, :] + np.eye(bs) * reg
        b = p @ gram + reg * p[bb:bb + bs]
        for k, (c, v) in enumerate(zip(keys, vals)):
            b = b + (Y[k] - 1.0) * v * alpha * Qd[c, bb:bb + bs]
        x = np.zeros(bs); r = b.copy(); pv = r.copy(); rsold = r @ r
        if rsold > tol:
            for _ in range(3):
                Ap = A @ pv
                for c, v in zip(keys, vals):
                    qb = Qd[c, bb:bb + bs]
                    Ap = Ap + v * alpha * (qb @ pv) * qb
                step = rsold / (pv @ Ap)
                x = x + step * pv; r = r - step * Ap
                rsnew = r @ r
                if rsnew < tol: break
                pv = r + (rsnew / rsold) * pv; rsold = rsnew
        p[bb:bb + bs] -= x
        for k, c in enumerate(keys):
            Y[k] -= Qd[c, bb:bb + bs] @ x
    return p


def cg_f64(x0, A, y, iters, tol=1e-10, eps=1e-10):
    x = x0.astype(np.float64).copy(); r = y - x @ A
    if y @ y < r @ r:
        x[:] = 0; r = y.copy()
    p = r.copy(); rs = r @ r
    for _ in range(iters):
        Ap = p @ A; a = rs / (Ap @ p + eps); x = x + a * p; r = r - a * Ap
        rn = r @ r
        if rn < tol: break
        p = r + rn / (rs + eps) * p; rs = rn
    return x


for d, kw in ((70, dict(optimizer="manual_cg", adaptive_reg=True, num_cg_max_iters=5)), (128, dict(optimizer="manual_cg")),
              (256, dict(optimizer="manual_cg")), (20, dict(optimizer="manual_cg"))):
    csr = tiny_csr(U=320, I=280, density=0.06, seed=31, counts=True)
    opt = als_opt(d=d, alpha=4.0, reg_u=0.2, reg_i=0.3, num_iters=2, **kw)
    vdim = ((d + 31) // 32) * 32
    rng = np.random.default_rng(3)
    P = H.pad(np.abs(rng.normal(scale=0.1, size=(320, d))).astype(np.float32), vdim)
    Q = H.pad(np.abs(rng.normal(scale=0.1, size=(280, d))).astype(np.float32), vdim)
    P0, Q0 = P[:, :d].copy(), Q[:, :d].copy()
    Po, Qo = P0.copy(), Q0.copy()
    o = orc.OracleALS(); assert o.init(H.write_opt(opt)); o.initialize_model(Po, Qo)
    g = CyALS(); assert g.init(H.write_opt(dict(opt, accelerator=True))); g.initialize_model(P, Q)
    t = csr.transpose(); g.set_placeholder(csr.indptr, t.indptr, csr.nnz + 1)
    o.precompute(0); g.precompute(0)
    FF = o.get_ff(d)
    o.partial_update(0, 320, csr.indptr, csr.keys, csr.vals, 0)
    g.partial_update(0, 320, csr.indptr, csr.keys, csr.vals, 0)
    T = np.zeros((320, d))
    for u in range(320):
        keys, vals = csr.row(u)
        if d >= 128:
            T[u] = ialspp_f64(P0, Q0, FF, u, keys, vals, 4.0, 0.2, 32)
        else:
            A = FF.astype(np.float64).copy(); y = np.zeros(d)
            for c, v in zip(keys, vals):
                q = Q0[c].astype(np.float64); A += 4.0 * v * np.outer(q, q); y += q * (1 + v * 4.0)
            ada = len(keys) if opt["adaptive_reg"] else 1.0
            A += np.eye(d) * 0.2 * ada
            T[u] = cg_f64(P0[u], A, y, opt["num_cg_max_iters"])
    eo, eg, eog = H.relerr(Po, T), H.relerr(P[:, :d], T), H.relerr(P[:, :d], Po)
    rows_o = np.abs(Po - T).max(1) / np.abs(T).max(); rows_g = np.abs(P[:, :d] - T).max(1) / np.abs(T).max()
    print("d=%d %s: oracle-vs-f64 %.2e  hip-vs-f64 %.2e  hip-vs-oracle %.2e | worst rows oracle %s hip %s deg %s"
          % (d, kw, eo, eg, eog, np.argsort(-rows_o)[:3], np.argsort(-rows_g)[:3],
             [len(csr.row(int(u))[0]) for u in np.argsort(-rows_g)[:3]]))
