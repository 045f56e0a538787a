"""Which user rows of the config-#3 warm epoch sit furthest from float64, per path (the oracle = reference path on the host cores,
fp32 matrix instruction, split-f16): row, length, error.  Diagnostic (the oracle is the checker here, as in tests/).
    python scripts/als_split_rows.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import ref_numpy as rn
from buffalo_amd import ingest, synth
from buffalo_amd.backend import CyALS

csr = bench.load_matrix("ml20m", 7)
U, I, nnz = csr.num_users, csr.num_items, csr.nnz
vals = (1 + np.random.default_rng(7).poisson(1.0, size=nnz)).astype(np.float32)
col = ingest.coo_to_csr(csr.keys, csr.rows(), vals, I, U)
D, OPT = bench.D, bench.ALS_OPT


def make(P, Q, modes):
    g = CyALS()
    path = bench._opt_file(OPT)
    assert g.init(path)
    os.unlink(path)
    g.initialize_model(P, Q)
    g.set_resident_csr(0, csr.indptr, csr.keys, vals)
    g.set_resident_csr(1, col["indptr"], col["key"], col["val"])
    for k, v in modes.items():
        g.set_mode(k, v)
    return g


def half(g, axis):
    rows, ip = (U, csr.indptr) if axis == 0 else (I, col["indptr"])
    g.precompute(axis)
    g.partial_update(0, rows, ip, None, None, axis)


from oracle import oracle as orc
orc.build()
import helpers as H
for warm_mode in (1,):
    P, Q, _ = synth.init_factors(U, I, D, seed=7)
    g = make(P, Q, {"als_split_f16": warm_mode})
    for ep in range(2):
        half(g, 0), half(g, 1)
    Pw, Qw = P.copy(), Q.copy()
    del g
    res, ff = {}, {}
    for name, m in (("fp32", 0), ("split", 1)):
        P1, Q1 = Pw.copy(), Qw.copy()
        g = make(P1, Q1, {"als_split_f16": m})
        g.precompute(0)
        ff[name] = g.device_tensor("FF", (D, D)).cpu().numpy().astype(np.float64)
        g.partial_update(0, U, csr.indptr, None, None, 0)
        res[name] = P1.copy()
        del g
    Po, Qo = Pw.copy(), Qw.copy()
    o = orc.OracleALS()
    assert o.init(H.write_opt(dict(OPT, accelerator=False, num_workers=os.cpu_count() or 16)))
    o.initialize_model(Po, Qo)
    o.precompute(0)
    ff["oracle"] = o.get_ff(D).astype(np.float64)
    o.partial_update(0, U, csr.indptr, csr.keys, vals, 0)
    res["oracle"] = Po
    ip = np.concatenate([[0], csr.indptr])
    scale = float(np.abs(res["fp32"]).max())
    for a in (0, 20000, 40000, 68996, 100000, 120000, 137993):
        rows = []
        for u in range(a, a + 500):
            b, e = int(ip[u]), int(ip[u + 1])
            if e == b:
                continue
            errs = []
            for name in ("fp32", "split", "oracle"):
                t = rn.ialspp_row_f64_fast(Pw[u], Qw[csr.keys[b:e]], ff[name], vals[b:e], OPT["alpha"], OPT["reg_u"], OPT["block_size"])
                errs.append(float(np.abs(res[name][u] - t).max()) / scale)
            rows.append((max(errs), errs[0], errs[1], errs[2], u, e - b))
        rows.sort(reverse=True)
        print("warm state from the %s path, user rows [%d, %d): max err fp32 %.2e  split %.2e  oracle %.2e;  worst rows (row, n, fp32, split, oracle): %s" % (
            "split" if warm_mode else "fp32", a, a + 500, max(r[1] for r in rows), max(r[2] for r in rows), max(r[3] for r in rows),
            ", ".join("(%d, %d, %.1e, %.1e, %.1e)" % (r[4], r[5], r[1], r[2], r[3]) for r in rows[:6])), flush=True)
