"""CFR (CoFactor) parity: HIP backend vs the oracle's restatement of CCFR (lib/algo_impl/cfr/cfr.cc), epoch by epoch
in the order CFR._iterate calls them (cfr.py: precompute(item) -> users, precompute(user) -> items, contexts).

Tolerance: both sides solve the same d x d systems in fp32 (llt / ldlt exactly, manual_cg in 3 warm-started steps);
the systems are built in different summation orders, so factors agree to 2e-4 relative to the largest entry (1e-3 at
d = 96, where 150 users / 90 items leave the Gramians rank-deficient and the regulariser carries the solve, and for
the 4500-entry rows of the "long" case) and the
losses to 1e-4 -- the reference's own tests only check that CFR trains (tests/algo/test_cfr.py)."""
import numpy as np
import pytest

import helpers as H
from conftest import tiny_csr

pytestmark = pytest.mark.gpu


def _opt(**kw):
    opt = {"d": 20, "num_workers": 2, "num_cg_max_iters": 3, "alpha": 4.0, "l": 0.7, "eps": 1e-10, "reg_u": 0.1, "reg_i": 0.2, "reg_c": 0.3,
           "compute_loss": True, "optimizer": "llt", "cg_tolerance": 1e-10, "num_iters": 2, "model_path": "", "data_opt": {}}
    opt.update(kw)
    return opt


def _arrays(Uu, Ii, d, seed):
    rng = np.random.default_rng(seed)
    f = lambda r, c: rng.normal(scale=0.2, size=(r, c)).astype(np.float32)     # noqa: E731
    return {"user": f(Uu, d), "item": f(Ii, d), "context": f(Ii, d), "item_bias": f(Ii, 1), "context_bias": f(Ii, 1)}


def _bind(obj, arrs):
    for name in ("user", "item", "context", "item_bias", "context_bias"):
        obj.set_embedding(arrs[name], name.encode("utf8"))


def _epoch(obj, csr, t, ctx, n_chunks=1):
    losses = []
    obj.precompute(b"item")
    tot = 0.0
    for a, b in H.chunks_of(csr, n_chunks):
        keys, vals = H.chunk_arrays(csr, a, b)
        tot += obj.partial_update_user(a, b, csr.indptr, keys, vals)
    losses.append(tot)
    obj.precompute(b"user")
    tot = 0.0
    for a, b in H.chunks_of(t, n_chunks):
        ku, vu = H.chunk_arrays(t, a, b)
        kc, vc = H.chunk_arrays(ctx, a, b)
        tot += obj.partial_update_item(a, b, t.indptr, ku, vu, ctx.indptr, kc, vc)
    losses.append(tot)
    tot = 0.0
    for a, b in H.chunks_of(ctx, n_chunks):
        keys, vals = H.chunk_arrays(ctx, a, b)
        tot += obj.partial_update_context(a, b, ctx.indptr, keys, vals)
    losses.append(tot)
    return losses


@pytest.mark.parametrize("d,kw,shape", [(20, dict(optimizer="llt"), "tiny"), (20, dict(optimizer="manual_cg"), "tiny"),
                                        (40, dict(optimizer="ldlt", l=1.0), "tiny"), (96, dict(optimizer="llt", compute_loss=False), "tiny"),
                                        (32, dict(optimizer="manual_cg"), "long")])
def test_epochs_match_oracle(oracle, d, kw, shape):
    from buffalo_amd.backend import CyCFR
    if shape == "tiny":
        Uu, Ii = 150, 90
        csr = tiny_csr(U=Uu, I=Ii, density=0.1, seed=3, counts=True)
        ctx = tiny_csr(U=Ii, I=Ii, density=0.15, seed=4, counts=True)      # SPPMI-like item x item matrix
        ctx.indptr[40:] -= 0                                               # (every row non-empty by construction)
    else:
        Uu, Ii = 5000, 60                                                   # item rows of ~4500 users: chunks of a heavy row add into one slot
        csr = tiny_csr(U=Uu, I=Ii, density=0.9, seed=5, counts=True)
        ctx = tiny_csr(U=Ii, I=Ii, density=0.5, seed=6, counts=True)
    t = csr.transpose()
    opt = _opt(d=d, **kw)
    A, B = _arrays(Uu, Ii, d, seed=9), _arrays(Uu, Ii, d, seed=9)
    o = oracle.OracleCFR()
    assert o.init(H.write_opt(opt))
    _bind(o, A)
    g = CyCFR()
    assert g.init(H.write_opt(opt))
    _bind(g, B)
    for it in range(2):
        lo = _epoch(o, csr, t, ctx, n_chunks=1)
        lg = _epoch(g, csr, t, ctx, n_chunks=2 if it == 0 else 1)
        for name in A:
            assert H.relerr(B[name], A[name]) < (2e-4 if d < 64 and shape == "tiny" else 1e-3), (it, name, H.relerr(B[name], A[name]))
        for a, b in zip(lo, lg):
            assert abs(a - b) <= 1e-4 * max(1.0, abs(a)), (it, lo, lg)
        if not opt["compute_loss"]:
            assert lg == [0.0, 0.0, 0.0]
        for name in A:                                     # re-synchronise so differences do not compound
            B[name][...] = A[name]
        _bind(g, B)


def test_rows_without_entries_are_left_alone(oracle):
    """cfr.cc:112-115, 177-180, 284-287: empty rows are skipped; an item with context entries only is still solved."""
    from buffalo_amd.backend import CyCFR
    from buffalo_amd.synth import CSR
    Uu, Ii, d = 6, 5, 8
    csr = CSR(Uu, Ii, [2, 2, 3, 3, 5, 6], [0, 3, 1, 0, 4, 2], np.array([1, 2, 1, 3, 1, 2], np.float32))     # users 1 and 3 are empty
    t = csr.transpose()
    ctx = CSR(Ii, Ii, [1, 1, 3, 4, 4], [2, 0, 4, 1], np.array([0.5, 1.5, 0.25, 2.0], np.float32))           # items 1 and 4 have no context
    opt = _opt(d=d, optimizer="llt")
    A, B = _arrays(Uu, Ii, d, seed=2), _arrays(Uu, Ii, d, seed=2)
    before = {k: v.copy() for k, v in B.items()}
    o, g = oracle.OracleCFR(), CyCFR()
    assert o.init(H.write_opt(opt)) and g.init(H.write_opt(opt))
    _bind(o, A), _bind(g, B)
    _epoch(o, csr, t, ctx), _epoch(g, csr, t, ctx)
    for name in A:
        assert H.relerr(B[name], A[name]) < 2e-4, name
    assert np.array_equal(B["user"][1], before["user"][1]) and np.array_equal(B["user"][3], before["user"][3])
    assert np.array_equal(B["context"][1], before["context"][1]) and np.array_equal(B["context_bias"][4], before["context_bias"][4])
    assert not np.array_equal(B["item"][1], before["item"][1])          # user entries only: solved, bias becomes 0 / 1e-10 = 0
    assert B["item_bias"][1, 0] == 0.0 == A["item_bias"][1, 0]


def test_stream_to_sppmi_to_cfr_epoch(oracle):
    """The whole data path of CoFactor on the device: a stream of user sequences -> rowwise matrix (bfh_coo_to_csr) and SPPMI
    context matrix (bfh_sppmi_*, windows 3) -> one CFR epoch (bfh_cfr_*), against the same chain on the oracle
    (stream.py:240-267 + fileio.hpp:109-254 -> cfr.cc)."""
    from collections import Counter
    from buffalo_amd import ingest
    from buffalo_amd.backend import CyCFR
    from buffalo_amd.synth import CSR
    Uu, Ii, d = 400, 120, 24
    rng = np.random.default_rng(12)
    lens = rng.integers(2, 40, size=Uu)
    indptr = np.cumsum(lens).astype(np.int64)
    pop = 1.0 / np.arange(1, Ii + 1)
    items = rng.choice(Ii, size=int(indptr[-1]), p=pop / pop.sum()).astype(np.int32)
    # internal_data_type "matrix": per-user counts of the sequence (stream.py:252-255)
    rows, cols, vals = [], [], []
    beg = 0
    for u, end in enumerate(indptr):
        for c, v in sorted(Counter(items[beg:end].tolist()).items()):
            rows.append(u), cols.append(c), vals.append(float(v))
        beg = int(end)
    rows, cols, vals = np.array(rows, np.int32), np.array(cols, np.int32), np.array(vals, np.float32)
    g_row = ingest.coo_to_csr(rows, cols, vals, Uu, Ii)
    o_row = oracle.coo_to_csr(rows, cols, vals, Uu, Ii)
    g_sp = ingest.build_sppmi(indptr, items, Ii, 3, 1)
    o_sp = oracle.build_sppmi(indptr, items, Ii, 3, 1)
    for k in ("indptr", "key"):
        assert np.array_equal(g_row[k], o_row[k]) and np.array_equal(g_sp[k], o_sp[k])
    assert np.array_equal(g_sp["val"].view(np.int32), o_sp["val"].view(np.int32)) and len(g_sp["key"]) > Ii
    opt = _opt(d=d, optimizer="llt")
    A, B = _arrays(Uu, Ii, d, seed=4), _arrays(Uu, Ii, d, seed=4)
    o, g = oracle.OracleCFR(), CyCFR()
    assert o.init(H.write_opt(opt)) and g.init(H.write_opt(opt))
    _bind(o, A), _bind(g, B)
    mk = lambda grp, r, c: CSR(r, c, grp["indptr"], grp["key"], grp["val"])     # noqa: E731
    csr_o, ctx_o = mk(o_row, Uu, Ii), mk(o_sp, Ii, Ii)
    csr_g, ctx_g = mk(g_row, Uu, Ii), mk(g_sp, Ii, Ii)
    lo = _epoch(o, csr_o, csr_o.transpose(), ctx_o)
    lg = _epoch(g, csr_g, csr_g.transpose(), ctx_g)
    for name in A:
        assert H.relerr(B[name], A[name]) < 2e-4, (name, H.relerr(B[name], A[name]))
    for a, b in zip(lo, lg):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(a)), (lo, lg)


def test_unsupported_options_fail_loudly():
    from buffalo_amd._lib import BuffaloHipError
    from buffalo_amd.backend import CyCFR
    with pytest.raises(BuffaloHipError):
        CyCFR().init(H.write_opt(_opt(optimizer="eigen_cg")))
    with pytest.raises(BuffaloHipError):
        CyCFR().init(H.write_opt(_opt(d=160)))
    g = CyCFR()
    assert g.init(H.write_opt(_opt()))
    with pytest.raises(BuffaloHipError):
        g.precompute(b"item")
    with pytest.raises(BuffaloHipError):
        g.set_embedding(np.zeros((3, 20), np.float32), b"nonsense")
