"""eALS parity: HIP backend vs the oracle's restatement of CEALS (lib/algo_impl/eals/eals.cc), in the order
EALS._iterate drives it (eals.py:62-80: precompute both caches once, then update(axis 0), update(axis 1) per epoch).

Coordinate descent is sequential in d and every step divides two sums over the row's entries; the two sides add
them in different orders (lane partials + wave reduction vs a running float), so factors agree to 1e-4 relative to
the largest entry after each half-epoch (the backends are re-synchronised in between) and to 2e-3 when five epochs
run free.  The oracle itself is pinned in tests/test_oracle_pins.py (its loss equals the float64 objective and
falls monotonically)."""
import numpy as np
import pytest

import helpers as H
from conftest import tiny_csr

pytestmark = pytest.mark.gpu


def _opt(d, **kw):
    opt = {"d": d, "num_workers": 2, "alpha": 2.0, "reg_u": 0.1, "reg_i": 0.2, "num_iters": 3, "c0": 0.5, "exponent": 0.5, "model_path": "",
           "data_opt": {}}
    opt.update(kw)
    return opt


def _problem(shape, d, seed=0):
    if shape == "tiny":
        csr = tiny_csr(U=120, I=70, density=0.12, seed=1, counts=True)
    elif shape == "empty_rows":
        from buffalo_amd.synth import CSR
        csr = CSR(6, 5, [2, 2, 3, 3, 5, 6], [0, 3, 1, 0, 4, 2], np.array([1, 2, 1, 3, 1, 2], np.float32))
    elif shape == "long":   # rows beyond 256 entries take the streaming path: item rows of ~540, user rows of ~54
        csr = tiny_csr(U=600, I=60, density=0.9, seed=2, counts=True)
    else:   # "heavy": item rows of ~2300 entries run on the block-per-row kernel (> 1024 entries)
        csr = tiny_csr(U=2500, I=24, density=0.92, seed=4, counts=True)
    rng = np.random.default_rng(seed)
    P = rng.normal(scale=0.3, size=(csr.num_users, d)).astype(np.float32)
    Q = rng.normal(scale=0.3, size=(csr.num_items, d)).astype(np.float32)
    pop = np.bincount(csr.keys, minlength=csr.num_items).astype(np.float64) ** 0.5      # eals.py:50-53
    Cw = (0.5 * pop / pop.sum()).astype(np.float32)
    return csr, P, Q, Cw


def _pair(oracle, opt, csr, P, Q, Cw):
    from buffalo_amd.backend import CyEALS
    t = csr.transpose()
    Po, Qo = P.copy(), Q.copy()
    o = oracle.OracleEALS()
    assert o.init(H.write_opt(opt))
    o.initialize_model(Po, Qo, Cw)
    g = CyEALS()
    assert g.init(H.write_opt(opt))
    g.initialize_model(P, Q, Cw)
    assert not o.update(csr.indptr, csr.keys, csr.vals, 0) and not g.update(csr.indptr, csr.keys, csr.vals, 0)    # eals.cc:106-114
    assert g.estimate_loss(csr.nnz, csr.indptr, csr.keys, csr.vals, 0) == (0.0, 0.0)
    for obj in (o, g):
        obj.precompute_cache(csr.nnz, csr.indptr, csr.keys, 0)
        obj.precompute_cache(csr.nnz, t.indptr, t.keys, 1)
    return o, g, t, Po, Qo


@pytest.mark.parametrize("d,shape", [(20, "tiny"), (128, "tiny"), (20, "empty_rows"), (40, "long"), (24, "heavy")])
def test_half_epochs_match_oracle(oracle, d, shape):
    csr, P, Q, Cw = _problem(shape, d)
    opt = _opt(d)
    o, g, t, Po, Qo = _pair(oracle, opt, csr, P, Q, Cw)
    lo, lg = o.estimate_loss(csr.nnz, csr.indptr, csr.keys, csr.vals, 0), g.estimate_loss(csr.nnz, csr.indptr, csr.keys, csr.vals, 0)
    assert abs(lo[0] - lg[0]) <= 1e-5 * max(1.0, lo[0]) and abs(lo[1] - lg[1]) <= 1e-4 * max(1.0, abs(lo[1]))
    before = P.copy()
    for it in range(2):
        for axis, m in ((0, csr), (1, t)):
            assert o.update(m.indptr, m.keys, m.vals, axis) and g.update(m.indptr, m.keys, m.vals, axis)
            X, Xo = (P, Po) if axis == 0 else (Q, Qo)
            assert H.relerr(X, Xo) < 1e-4, (it, axis, H.relerr(X, Xo))
            lo, lg = o.estimate_loss(csr.nnz, m.indptr, m.keys, m.vals, axis), g.estimate_loss(csr.nnz, m.indptr, m.keys, m.vals, axis)
            assert abs(lo[0] - lg[0]) <= 1e-4 * max(1.0, lo[0]) and abs(lo[1] - lg[1]) <= 2e-4 * max(1.0, abs(lo[1])), (lo, lg)
    if shape == "empty_rows":   # users 1 and 3 have no entries: the closed form still runs on the regulariser (eals.cc:193-216)
        assert np.array_equal(P[1] == before[1], Po[1] == before[1])
    assert np.isfinite(P).all() and np.isfinite(Q).all()


def test_free_running_epochs_stay_close_and_descend(oracle):
    csr, P, Q, Cw = _problem("tiny", 32, seed=3)
    o, g, t, Po, Qo = _pair(oracle, _opt(32), csr, P, Q, Cw)
    losses = [g.estimate_loss(csr.nnz, csr.indptr, csr.keys, csr.vals, 0)[1]]
    for _ in range(5):
        for axis, m in ((0, csr), (1, t)):
            o.update(m.indptr, m.keys, m.vals, axis)
            g.update(m.indptr, m.keys, m.vals, axis)
        losses.append(g.estimate_loss(csr.nnz, csr.indptr, csr.keys, csr.vals, 0)[1])
    assert all(b <= a * (1 + 1e-5) for a, b in zip(losses, losses[1:])), losses      # exact coordinate minimisation never goes up
    assert losses[-1] < 0.5 * losses[0]
    assert H.relerr(P, Po) < 2e-3 and H.relerr(Q, Qo) < 2e-3
    # the cache stays consistent with the factors: loss from the cache == objective recomputed from P, Q in float64
    rows = np.repeat(np.arange(csr.num_users), np.diff(np.concatenate([[0], csr.indptr])))
    P64, Q64 = P.astype(np.float64), Q.astype(np.float64)
    S = P64 @ Q64.T
    W = np.tile(Cw.astype(np.float64), (csr.num_users, 1))
    R = np.zeros_like(S)
    R[rows, csr.keys] = csr.vals
    W[rows, csr.keys] = 1 + 2.0 * csr.vals
    want = (W * (R - S) ** 2).sum() + 0.1 * (P64 ** 2).sum() + 0.2 * (Q64 ** 2).sum()
    assert abs(losses[-1] - want) < 2e-4 * want
