"""Whole-run ALS parity, pair by pair and epoch by epoch: HIP vs oracle vs a float64 evaluation of the same recurrence
(tests/ref_numpy.py: als.cc:107-209 + algo.cc:58-82 in double).  TEST/STUDY TOOL, not on the product path.

Two views per half-epoch:
  * free-running: each of the three keeps ITS OWN factors from epoch to epoch (what a whole-run comparison sees);
  * one-step: all three start the half-epoch from the ORACLE's current factors (what the kernels themselves add).
If err(hip, f64) <= ~err(oracle, f64) in both views, a whole-run gap hip~oracle is the conditioning of the case, not a defect.

    python scripts/als_cg_diag.py [case ...]      # cases: cg64_90 (the round-2 failure), cg64_2000 (well-posed), llt32_90
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import helpers as H            # noqa: E402
import ref_numpy as rn         # noqa: E402
from conftest import als_opt   # noqa: E402


def planted(U, I, seed, p_in=0.45, p_out=0.04):
    """tests/golden/make_trained_models.coordinate_text's matrix, as CSR."""
    from buffalo_amd.synth import CSR
    rng = np.random.default_rng(seed)
    ug, ig = rng.integers(0, 6, U), rng.integers(0, 6, I)
    p = np.where(ug[:, None] == ig[None, :], p_in, p_out)
    M = rng.random((U, I)) < p
    M[np.arange(U), rng.integers(0, I, size=U)] = True
    M[rng.integers(0, U, size=I), np.arange(I)] = True
    r, c = np.nonzero(M)
    vals = rng.integers(1, 6, len(r)).astype(np.float32)
    return CSR(U, I, np.cumsum(np.bincount(r, minlength=U), dtype=np.int64), c.astype(np.int32), vals)


CASES = {
    "cg64_90": ((150, 90, 2, 0.45, 0.04), dict(d=64, num_iters=4, optimizer="manual_cg", alpha=4.0, reg_u=0.05, reg_i=0.2)),
    "cg64_2000": ((3000, 2000, 2, 0.10, 0.01), dict(d=64, num_iters=4, optimizer="manual_cg", alpha=4.0, reg_u=0.05, reg_i=0.2)),
    "llt32_90": ((150, 90, 1, 0.45, 0.04), dict(d=32, num_iters=3, optimizer="llt")),
    "llt64_2000": ((3000, 2000, 2, 0.10, 0.01), dict(d=64, num_iters=3, optimizer="llt", alpha=4.0, reg_u=0.05, reg_i=0.2)),
}


def run(name):
    from buffalo_amd.backend import CyALS
    from oracle import oracle as orc
    shape, over = CASES[name]
    csr = planted(*shape)
    t = csr.transpose()
    opt = als_opt(compute_loss_on_training=False, **over)
    d = opt["d"]
    rng = np.random.default_rng(9)
    P0 = np.abs(rng.normal(scale=1.0 / d, size=(csr.num_users, d))).astype(np.float32)
    Q0 = np.abs(rng.normal(scale=1.0 / d, size=(csr.num_items, d))).astype(np.float32)
    Po, Qo = P0.copy(), Q0.copy()          # oracle, free-running
    Pg, Qg = P0.copy(), Q0.copy()          # HIP, free-running
    P6, Q6 = P0.astype(np.float64), Q0.astype(np.float64)   # float64, free-running
    o = orc.OracleALS()
    assert o.init(H.write_opt(opt))
    o.initialize_model(Po, Qo)
    g = CyALS()
    assert g.init(H.write_opt(dict(opt, accelerator=True)))
    g.initialize_model(Pg, Qg)
    g.set_placeholder(csr.indptr, t.indptr, csr.nnz + 1)
    # second HIP object for the one-step view
    Ps, Qs = P0.copy(), Q0.copy()
    print("== %s: %d x %d, nnz %d, d %d, %s" % (name, csr.num_users, csr.num_items, csr.nnz, d, opt["optimizer"]))
    print("%-9s | free-running: hip~f64  orc~f64  hip~orc | one-step from oracle state: hip~f64  orc~f64  hip~orc" % "epoch/ax")
    worst = dict(free=0.0, ratio_free=0.0, ratio_step=0.0)
    for it in range(opt["num_iters"]):
        for axis, mat in ((0, csr), (1, t)):
            # ---- one-step view: everything from the oracle's current factors
            Xo_in, Yo_in = (Po.copy(), Qo.copy()) if axis == 0 else (Qo.copy(), Po.copy())
            ff64 = Yo_in.astype(np.float64).T @ Yo_in.astype(np.float64)
            step64 = rn.als_half_epoch_f64(Xo_in.copy(), Yo_in, ff64, mat, opt, axis)
            Ps[:], Qs[:] = Po, Qo
            s = CyALS()
            assert s.init(H.write_opt(dict(opt, accelerator=True)))
            s.initialize_model(Ps, Qs)
            s.set_placeholder(csr.indptr, t.indptr, csr.nnz + 1)
            s.precompute(axis)
            s.partial_update(0, mat.num_users, mat.indptr, mat.keys, mat.vals, axis)
            step_hip = (Ps if axis == 0 else Qs).copy()
            # ---- free-running
            o.precompute(axis)
            g.precompute(axis)
            o.partial_update(0, mat.num_users, mat.indptr, mat.keys, mat.vals, axis)
            g.partial_update(0, mat.num_users, mat.indptr, mat.keys, mat.vals, axis)
            X6, Y6 = (P6, Q6) if axis == 0 else (Q6, P6)
            new6 = rn.als_half_epoch_f64(X6.copy(), Y6, Y6.T @ Y6, mat, opt, axis)
            X6[:] = new6
            Xo, Xg = (Po, Pg) if axis == 0 else (Qo, Qg)
            f = (H.relerr(Xg, X6), H.relerr(Xo, X6), H.relerr(Xg, Xo))
            st = (H.relerr(step_hip, step64), H.relerr(Xo, step64), H.relerr(step_hip, Xo))
            print("%d/%d       |               %.2e %.2e %.2e |                            %.2e %.2e %.2e"
                  % (it, axis, f[0], f[1], f[2], st[0], st[1], st[2]))
            worst["free"] = max(worst["free"], f[2])
            worst["ratio_free"] = max(worst["ratio_free"], f[0] / max(f[1], 1e-30))
            worst["ratio_step"] = max(worst["ratio_step"], st[0] / max(st[1], 5e-7))
    print("   worst hip~orc (free) %.3e; worst err(hip,f64)/err(orc,f64): free %.2f, one-step %.2f"
          % (worst["free"], worst["ratio_free"], worst["ratio_step"]))




# ------------------------------------------------------------------------------------------------------------------------
# front-level trace of the round-2 failure (tests/test_trained_models_ref.py[als_manual_cg_d64]): the same file, seeds and
# stand-in front as the test; every partial_update the front issues is replayed on the oracle and in float64 FROM THE SAME
# INPUTS (the host arrays right before the call), so a call that deviates is named, and the end state is compared with golden.
#     python scripts/als_cg_diag.py front [case] [--oracle-backend]     (--oracle-backend: self-test of this tracer on CPU)
# ------------------------------------------------------------------------------------------------------------------------
def front_trace(name="als_manual_cg_d64", oracle_backend=False):
    import json
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests", "front_harness"))
    import make_trained_models as mk
    import buffalo_front.algo as A
    import buffalo_front.algo.als as ha
    import buffalo_front.data as D
    from buffalo_front.data import MatrixMarketOptions
    from buffalo_amd.synth import CSR
    from oracle import oracle as orc
    golden = np.load(mk.OUT)
    algo, shape, over, np_seed = mk.CASES[name]
    d = over["d"]

    if oracle_backend:
        class Base(orc.OracleALS):
            def init(self, opt_path):
                return super().init(opt_path)

            def get_vdim(self):
                return d

            def set_placeholder(self, *a):
                pass
        D._group = lambda nr, nc, r, c, v: orc.coo_to_csr(r, c, v, nr, nc)
        import buffalo_front.algo.base as hb

        class OracleRanker:
            def dot_topn(self, rows, P, Q, Qb, keys, scores, pool, k):
                orc.dot_topn(np.ascontiguousarray(rows, dtype=np.int32), P, Q, Qb, keys, scores, pool, k)
        hb.Algo._ranker = lambda self: OracleRanker()
    else:
        from buffalo_amd.backend import CyALS as Base
    calls = []

    class Tracing(Base):
        def init(self, opt_path):
            p = opt_path.decode() if isinstance(opt_path, bytes) else opt_path
            self._opt = json.load(open(p))
            return super().init(opt_path)

        def initialize_model(self, P, Q):
            self._P, self._Q = P, Q
            return super().initialize_model(P, Q)

        def partial_update(self, start_x, next_x, indptr, keys, vals, axis):
            P0, Q0 = self._P[:, :d].copy(), self._Q[:, :d].copy()
            out = super().partial_update(start_x, next_x, indptr, keys, vals, axis)
            X = (self._P if axis == 0 else self._Q)[:, :d]
            opt = dict(self._opt, accelerator=False)
            o = orc.OracleALS()
            assert o.init(H.write_opt(opt))
            Po, Qo = P0.copy(), Q0.copy()
            o.initialize_model(Po, Qo)
            o.precompute(axis)
            out_o = o.partial_update(start_x, next_x, indptr, keys, vals, axis)
            Xo = Po if axis == 0 else Qo
            rows = X.shape[0]
            whole = start_x == 0 and next_x == rows
            e64 = (float("nan"), float("nan"))
            if whole:
                Y0 = Q0 if axis == 0 else P0
                X0 = P0 if axis == 0 else Q0
                mat = CSR(rows, Y0.shape[0], indptr, keys, vals)
                ff = Y0.astype(np.float64).T @ Y0.astype(np.float64)
                t = rn.als_half_epoch_f64(X0.copy(), Y0, ff, mat, opt, axis)
                e64 = (H.relerr(X, t), H.relerr(Xo, t))
                # row by row (rows of one half-epoch are independent solves): is the backend's error concentrated somewhere the oracle's is not?
                sc = max(np.abs(t).max(), 1e-30)
                rh, ro = np.abs(X - t).max(axis=1) / sc, np.abs(Xo - t).max(axis=1) / sc
                top = np.argsort(-rh)[:4]
                print("   call %d rows by err(hip,f64): %s | same rows err(orc,f64): %s | rms over rows hip %.2e orc %.2e | rows where hip > 3x orc: %d"
                      % (len(calls), " ".join("%d:%.1e" % (r, rh[r]) for r in top), " ".join("%.1e" % ro[r] for r in top),
                         float(np.sqrt((rh ** 2).mean())), float(np.sqrt((ro ** 2).mean())), int((rh > 3 * np.maximum(ro, 1e-7)).sum())))
                if opt["optimizer"] == "manual_cg" and d < 128:
                    # Q-17's discontinuity: warm start unless |y|^2 < |r|^2 -- how close is every row to the switch?
                    reg = opt["reg_u"] if axis == 0 else opt["reg_i"]
                    marg = []
                    for u in range(rows):
                        kk, vv = mat.row(u)
                        if len(kk) == 0:
                            continue
                        Am, y = rn.als_normal_equations(X0, Y0, ff, u, kk, vv, opt["alpha"], reg, opt["adaptive_reg"])
                        r = y - X0[u].astype(np.float64) @ Am
                        marg.append((float(y @ y) - float(r @ r)) / float(y @ y))
                    marg = np.asarray(marg)
                    print("   call %d: warm-start margin (|y|^2-|r|^2)/|y|^2: min %.3e  min|.| %.3e  cold rows %d of %d"
                          % (len(calls), marg.min(), np.abs(marg).min(), int((marg < 0).sum()), len(marg)))
            per_row = np.abs(X[start_x:next_x] - Xo[start_x:next_x]).max(axis=1) / max(np.abs(Xo).max(), 1e-30)
            calls.append((axis, start_x, next_x, len(keys), H.relerr(X[start_x:next_x], Xo[start_x:next_x]), e64, int((per_row > 1e-3).sum()),
                          int(per_row.argmax()) + start_x, out, out_o))
            return out

    ha.CyALS = Tracing
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "main.mtx")
        with open(path, "w") as f:
            f.write(mk.coordinate_text(*shape))
        opt = A.ALSOption().get_default_option()
        opt.update(over)
        np.random.seed(np_seed)
        model = A.ALS(opt, data_opt=mk.data_option(MatrixMarketOptions, path))
        model.initialize()
        # the training matrix the front feeds, against the oracle's builder on the same records
        g = model.data.get_group("rowwise")
        print("rowwise group: rows %d nnz %d  vals in [%g, %g]  empty rows %d" % (len(g["indptr"]), len(g["key"]), g["val"].min(), g["val"].max(),
              int((np.diff(np.concatenate([[0], g["indptr"]])) == 0).sum())))
        gc = model.data.get_group("colwise")
        print("colwise group: rows %d nnz %d  empty rows %d" % (len(gc["indptr"]), len(gc["key"]), int((np.diff(np.concatenate([[0], gc["indptr"]])) == 0).sum())))
        P_init, Q_init = model.P.copy(), model.Q.copy()
        ret = model.train()
    print("call axis rows          nnz   hip~orc(one step)  hip~f64   orc~f64  rows>1e-3 worst  loss hip / oracle")
    for i, c in enumerate(calls):
        print("%3d  %d   [%4d,%4d) %6d   %.3e         %.2e  %.2e  %4d     %4d   %s / %s"
              % (i, c[0], c[1], c[2], c[3], c[4], c[5][0], c[5][1], c[6], c[7], "(%.6g, %.6g)" % tuple(c[8]), "(%.6g, %.6g)" % tuple(c[9])))
    for f in ("P", "Q"):
        got, ref = getattr(model, f), golden["%s/%s" % (name, f)]
        print("end state vs golden %s: max-abs %.4e of max %.4f  (relerr %.3e)" % (f, np.abs(got - ref).max(), np.abs(ref).max(), H.relerr(got, ref)))
    print("train_loss %.8g   golden %.8g" % (ret["train_loss"], json.loads(str(golden["meta"]))[name]["train"]["train_loss"]))
    print("init factors: P sum %.9g  Q sum %.9g  (shape %s %s)" % (P_init.astype(np.float64).sum(), Q_init.astype(np.float64).sum(), P_init.shape, Q_init.shape))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "front":
        names = [a for a in sys.argv[2:] if not a.startswith("--")] or ["als_manual_cg_d64"]
        for n in names:
            front_trace(n, oracle_backend="--oracle-backend" in sys.argv)
    else:
        for n in (sys.argv[1:] or list(CASES)):
            run(n)
