"""A/B of the config-#3 ALS row kernel: fp32 matrix instruction (als_split_f16=0) against the split-f16 pass (als_split_f16=1).
  * kernel time per half-epoch of each;
  * from one common warm state (two epochs), one epoch with each: how far the two results are from each other and, on sampled
    rows, from the float64 evaluation of the same recurrence (tests/ref_numpy.py) -- the error each path adds on its own.
    python scripts/als_split_ab.py"""
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
D = bench.D
OPT = bench.ALS_OPT


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
    g.reset_stats()
    g.partial_update(0, rows, ip, None, None, axis)
    return g.stats()["kernel_ms"]


def timing(modes, epochs=4):
    P, Q, _ = synth.init_factors(U, I, D, seed=7)
    g = make(P, Q, dict(modes, als_writeback=0))
    per = {0: [], 1: []}
    for ep in range(epochs):
        for axis in (0, 1):
            ms = half(g, axis)
            if ep:
                per[axis].append(ms)
    print("%-28s user half-epoch %.3f ms  item half-epoch %.3f ms  sum %.3f" % (modes, np.mean(per[0]), np.mean(per[1]), np.mean(per[0]) + np.mean(per[1])), flush=True)


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


for m in ({"als_split_f16": 0}, {"als_split_f16": 1}):
    timing(m)

if os.environ.get("AB_TIMING_ONLY"):
    sys.exit(0)
# the common warm state
P, Q, _ = synth.init_factors(U, I, D, seed=7)
g = make(P, Q, {"als_split_f16": 0})
for ep in range(2):
    half(g, 0), half(g, 1)
Pw, Qw = P.copy(), Q.copy()
del g
out = {}
for name, m in (("fp32", {"als_split_f16": 0}), ("split", {"als_split_f16": 1})):
    P1, Q1 = Pw.copy(), Qw.copy()
    g = make(P1, Q1, m)
    half(g, 0)
    Pmid = P1.copy()
    half(g, 1)
    out[name] = (Pmid, Q1.copy())
    del g
print("one epoch from the same warm state, split-f16 vs fp32 instruction: P max-rel %.3e   Q max-rel %.3e" % (rel(out["split"][0], out["fp32"][0]), rel(out["split"][1], out["fp32"][1])), flush=True)
# float64 on sampled rows.  user half-epoch: inputs (Pw, Qw); item half-epoch of each path: inputs (its own Pmid, Qw)
rng = np.random.default_rng(3)
FFu = Qw.astype(np.float64).T @ Qw.astype(np.float64)
us = rng.choice(U, 400, replace=False)
ip = np.concatenate([[0], csr.indptr])
eu = {k: 0.0 for k in out}
for u in us:
    b, e = int(ip[u]), int(ip[u + 1])
    if e == b:
        continue
    t = rn.ialspp_row_f64_fast(Pw[u], Qw[csr.keys[b:e]], FFu, vals[b:e], OPT["alpha"], OPT["reg_u"], OPT["block_size"])
    for k in out:
        eu[k] = max(eu[k], float(np.abs(out[k][0][u] - t).max()))
scale = float(np.abs(out["fp32"][0]).max())
print("user rows (400 sampled) vs float64: fp32 instruction %.3e   split-f16 %.3e   (of the largest entry)" % (eu["fp32"] / scale, eu["split"] / scale), flush=True)
ipc = np.concatenate([[0], col["indptr"]])
its = rng.choice(I, 150, replace=False)
ei = {k: 0.0 for k in out}
for k in out:
    Pm = out[k][0]
    FFi = Pm.astype(np.float64).T @ Pm.astype(np.float64)
    for i in its:
        b, e = int(ipc[i]), int(ipc[i + 1])
        if e == b:
            continue
        t = rn.ialspp_row_f64_fast(Qw[i], Pm[col["key"][b:e]], FFi, col["val"][b:e], OPT["alpha"], OPT["reg_i"], OPT["block_size"])
        ei[k] = max(ei[k], float(np.abs(out[k][1][i] - t).max()))
scale = float(np.abs(out["fp32"][1]).max())
print("item rows (150 sampled) vs float64: fp32 instruction %.3e   split-f16 %.3e   (of the largest entry)" % (ei["fp32"] / scale, ei["split"] / scale), flush=True)
