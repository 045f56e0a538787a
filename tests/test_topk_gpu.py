"""Top-k selection parity: HIP backend (bfh_topk_* through buffalo_amd.parallel) vs the CPU oracle's
restatement of parallel::dot_topn / quickselect.

* exact-arithmetic cases (integer factors): keys AND scores bit-identical, incl. the admission rule, padding
  and the boundary-tie rule;
* float cases: the two backends sum the d products in different orders, so scores agree to 1e-6 relative to
  the row's largest |score| and keys agree wherever the oracle's neighbouring scores are further apart than that."""
import numpy as np
import pytest

import helpers as H
import topk_cases as tc

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["auto", "fused-c0-32", "fused-c0-rule"], autouse=True)
def fused_mode(request):
    """Every case runs three ways: the size rule (these shapes: the dense score buffer + select), the fused path forced
    with exactly one sampled tile (loose thresholds: long candidate lists, overflowing rows handed back to the dense path) and
    forced with the rule's sample size.  The fused path must be bit-identical to the dense one."""
    from buffalo_amd import parallel as par
    eng = par._engine()
    eng.set_mode("fused", -1 if request.param == "auto" else 1)
    eng.set_mode("fused_c0", 32 if request.param == "fused-c0-32" else 0)
    yield request.param
    eng.set_mode("fused", -1)
    eng.set_mode("fused_c0", 0)


def _check_float_case(oracle, par, indexes, P, Q, Qb, pool, k):
    gk, gs = tc.run(par.dot_topn, indexes, P, Q, Qb, pool, k)
    ok, os_ = tc.run(oracle.dot_topn, indexes, P, Q, Qb, pool, k)
    assert np.array_equal(gk == -1, ok == -1)
    scale = np.abs(os_).max(axis=1, keepdims=True) + 1e-30
    tol = 2e-6 * scale
    assert np.all(np.abs(gs - os_) <= tol), np.abs(gs - os_).max()
    # a key may differ only inside a run of oracle scores closer than the tolerance
    diff = gk != ok
    if diff.any():
        pad = np.pad(os_, ((0, 0), (1, 1)), mode="edge")
        near = (np.abs(pad[:, 1:-1] - pad[:, :-2]) <= 2 * tol) | (np.abs(pad[:, 1:-1] - pad[:, 2:]) <= 2 * tol)
        edge = np.zeros_like(diff)
        edge[:, -1] = True                      # the last admitted slot competes with the first rejected one
        assert np.all(~diff | near | edge), (gk[diff], ok[diff])
    valid = gk >= 0
    assert np.all(np.diff(np.where(valid, gs, -np.inf), axis=1)[valid[:, 1:]] <= 0)     # sorted by descending score
    return gk, gs


def test_reference_parallel_tests(oracle):
    """tests/parallel/test_base.py:38-101 (test01 most_similar, test03 pool, test04 topk) against numpy."""
    from buffalo_amd import parallel as par
    Q = tc.unit_factors(128, 5, seed=1)
    idx = np.arange(5, dtype=np.int32)
    keys, scores = tc.run(par.dot_topn, idx, Q, Q, tc.NO_BIAS, tc.EMPTY_POOL, 10)
    wk, ws = tc.numpy_most_similar(idx, Q, 10)
    assert np.allclose(keys, wk, atol=1e-7) and np.allclose(scores, ws, atol=1e-6)
    keys, scores = tc.run(par.dot_topn, idx, Q, Q, tc.NO_BIAS, np.array([5, 6, 7], np.int32), 10)
    assert set(keys.reshape(-1)) == {5, 6, 7, -1} and np.all(scores[:, 3:] == 0.0)
    P, Q2 = tc.unit_factors(512, 5, seed=3), tc.unit_factors(128, 5, seed=4)
    qi = np.array([312, 313, 314, 315, 316], dtype=np.int32)
    keys, scores = tc.run(par.dot_topn, qi, P, Q2, tc.NO_BIAS, tc.EMPTY_POOL, 10)
    wk, ws = tc.numpy_topk(qi, P, Q2, 10)
    assert np.allclose(keys, wk, atol=1e-7) and np.allclose(scores, ws, atol=1e-6)


@pytest.mark.parametrize("same,bias,pool,k", [(False, False, [], 7), (True, False, [], 7), (False, True, [], 40),
                                              (True, True, [1, 2, 3, 5, 8, 13, 21, 34], 5), (False, False, [0, 4], 6),
                                              (False, True, [], 1), (True, False, [], 37), (False, False, [], 300)])
def test_admission_padding_and_tie_rules_bit_exact(oracle, same, bias, pool, k):
    from buffalo_amd import parallel as par
    for rows, d, seed in ((37, 6, 5), (131, 12, 9), (1000, 40, 10)):
        Q = tc.integer_factors(rows, d, seed=seed)
        P = Q if same else tc.integer_factors(70, d, seed=seed + 1)
        Qb = tc.integer_factors(rows, 1, seed=seed + 2, lo=-1, hi=2) if bias else tc.NO_BIAS
        idx = np.arange(min(P.shape[0], 37), dtype=np.int32)[::-1].copy()
        gk, gs = tc.run(par.dot_topn, idx, P, Q, Qb, np.array(pool, np.int32), k)
        ok, os_ = tc.run(oracle.dot_topn, idx, P, Q, Qb, np.array(pool, np.int32), k)
        assert np.array_equal(gk, ok), (rows, d)
        assert np.array_equal(gs, os_), (rows, d)


@pytest.mark.parametrize("d,q_rows,nq,k", [(5, 128, 5, 10), (12, 1000, 33, 10), (32, 1682, 943, 50), (128, 3000, 257, 100),
                                           (200, 777, 129, 20), (256, 500, 64, 499)])
def test_float_factors_match_oracle(oracle, d, q_rows, nq, k):
    """d not a multiple of 8, d > 128 (two K-chunks through the score buffer), ragged tile edges."""
    from buffalo_amd import parallel as par
    rng = np.random.default_rng(d + q_rows)
    P = rng.normal(scale=0.3, size=(nq + 11, d)).astype(np.float32)
    Q = rng.normal(scale=0.3, size=(q_rows, d)).astype(np.float32)
    Qb = rng.normal(scale=0.1, size=(q_rows, 1)).astype(np.float32)
    idx = rng.permutation(nq + 11)[:nq].astype(np.int32)
    _check_float_case(oracle, par, idx, P, Q, tc.NO_BIAS, tc.EMPTY_POOL, k)
    _check_float_case(oracle, par, idx, P, Q, Qb, rng.permutation(q_rows)[: q_rows // 3].astype(np.int32), k)
    idx2 = idx[idx < q_rows]
    if len(idx2):
        _check_float_case(oracle, par, idx2, Q, Q, tc.NO_BIAS, tc.EMPTY_POOL, min(k, 64))   # most_similar: self excluded


@pytest.mark.parametrize("same,bias,pool_n,k,flt", [(False, False, 0, 10, 1), (False, True, 0, 100, 1), (True, False, 0, 25, 1),
                                                    (False, True, 700, 30, 1), (False, False, 0, 64, 0), (True, True, 40, 50, 0)])
def test_fused_path_is_bit_identical_to_dense(fused_mode, same, bias, pool_n, k, flt):
    """Own engine: 6,000 candidates x 300 queries, d=96: float factors with duplicated rows (exact ties at every rank),
    a constant block (hundreds of equal scores: ties straddling the k-th place and overflowing candidate lists) and an
    all-negative user (nothing admissible under the FLT_MIN rule)."""
    if fused_mode == "auto":
        pytest.skip("comparison of the two forced paths")
    from buffalo_amd import parallel as par
    rng = np.random.default_rng(k + pool_n)
    Q = rng.normal(scale=0.3, size=(6000, 96)).astype(np.float32)
    Q[1000:1400] = Q[2000:2400]                      # exact duplicates
    Q[3000:3300] = Q[3000]                           # one row 300 times
    P = Q if same else rng.normal(scale=0.3, size=(300, 96)).astype(np.float32)
    if not same:
        P[7] = -np.abs(P[7])
        Q[:, :48] = np.abs(Q[:, :48])                 # ... so that user 7's scores lean negative
        P[7, 48:] = 0.0
        P[9] = 4.0 * Q[3000]                          # the constant block is this user's best score
    Qb = rng.normal(scale=0.1, size=(6000, 1)).astype(np.float32) if bias else tc.NO_BIAS
    pool = rng.permutation(6000)[:pool_n].astype(np.int32) if pool_n else tc.EMPTY_POOL
    idx = np.arange(300, dtype=np.int32)
    out = {}
    for name, fused, c0 in (("dense", 0, 0), ("fused", 1, 0), ("fused32", 1, 32), ("fused1024", 1, 1024)):
        eng = par.TopK()
        eng.set_mode("fused", fused)
        eng.set_mode("fused_c0", c0)
        eng.set_mode("flt_min_rule", flt)
        out[name] = tc.run(lambda *a: eng.dot_topn(*a[:8]), idx, P, Q, Qb, pool, k)
        out[name + "_redo"] = eng.stats()["merges"]
    assert out["dense_redo"] == 0
    print("rows handed back to the dense path: c0 rule %d, c0=32 %d, c0=1024 %d of 300" % (out["fused_redo"], out["fused32_redo"], out["fused1024_redo"]))
    for name in ("fused", "fused32", "fused1024"):
        assert np.array_equal(out[name][0], out["dense"][0]), name
        assert np.array_equal(out[name][1], out["dense"][1]), name
    if pool_n == 0:
        assert out["fused1024_redo"] < 300 and out["fused_redo"] < 300     # the fused path itself produced rows ...
        assert out["fused32_redo"] > 0                                      # ... and a one-tile sample overflows lists: the dense redo path ran


def test_quickselect_matches_oracle(oracle):
    from buffalo_amd import parallel as par
    rng = np.random.default_rng(11)
    scores = (rng.permutation(77 * 1301).reshape(77, 1301) - 40000).astype(np.float32)   # distinct, both signs
    for k in (1, 10, 200, 1301):
        got, want = np.empty((77, k), np.int32), np.empty((77, k), np.int32)
        par.quickselect(scores, got, True)
        oracle.quickselect(scores, want, True)
        assert np.array_equal(got, want)
    got = np.empty((77, 25), np.int32)
    par.quickselect(scores, got, False)
    assert np.array_equal(np.sort(got, axis=1), np.sort(np.argsort(-scores, axis=1)[:, :25], axis=1))
    with pytest.raises(Exception):
        par.quickselect(scores, np.empty((77, 1302), np.int32), True)


def test_resident_factors_of_a_training_handle(oracle):
    """dot_topn_device ranks straight from the HBM buffers of CyBPR (P, Q, Qb) -- no factor upload."""
    from conftest import bpr_opt, tiny_csr
    from buffalo_amd import parallel as par
    from buffalo_amd.backend import CyBPR
    csr = tiny_csr(U=90, I=140, density=0.1, seed=2)
    d, vdim = 20, 32
    rng = np.random.default_rng(3)
    P = H.pad(rng.normal(scale=0.3, size=(90, d)).astype(np.float32), vdim)
    Q = H.pad(rng.normal(scale=0.3, size=(140, d)).astype(np.float32), vdim)
    Qb = rng.normal(scale=0.1, size=(140, 1)).astype(np.float32)
    obj = CyBPR()
    assert obj.init(H.write_opt(bpr_opt(d=d, accelerator=True)))
    obj.initialize_model(P, Q, Qb, csr.nnz, True)
    eng = par.TopK()
    idx = np.arange(90, dtype=np.int32)
    gk, gs = np.empty((90, 15), np.int32), np.empty((90, 15), np.float32)
    eng.dot_topn_device(idx, obj.device_buffer("P")[0], 90, obj.device_buffer("Q")[0], 140, d, vdim, obj.device_buffer("Qb")[0], False,
                        gk, gs, tc.EMPTY_POOL, 15)
    hk, hs = tc.run(par.dot_topn, idx, P, Q, Qb, tc.EMPTY_POOL, 15)
    # same kernels; the K-chunk is split between the half-waves at d_pad/2, which differs for d=20 (resident) and the
    # 32 padded columns the host call sees, so the sums may round differently
    assert np.abs(gs - hs).max() <= 2e-6 * np.abs(hs).max() and (gk == hk).mean() > 0.99
    ok, os_ = tc.run(oracle.dot_topn, idx, P, Q, Qb, tc.EMPTY_POOL, 15)
    assert np.abs(gs - os_).max() <= 2e-6 * np.abs(os_).max() and (gk == ok).mean() > 0.99


def test_par_classes_mirror_reference_surface(oracle):
    """ParALS / ParBPRMF (parallel/base.py:77-156; the harness stand-in) over a minimal algo object, ranking on the device."""
    from buffalo_front import parallel as par

    class Algo:
        pass
    rng = np.random.default_rng(8)
    algo = Algo()
    algo.P = rng.normal(size=(50, 16)).astype(np.float32)
    algo.Q = rng.normal(size=(80, 16)).astype(np.float32)
    algo.Qb = rng.normal(size=(80, 1)).astype(np.float32)
    keys, topks, scores = par.ParALS(algo).topk_recommendation(np.arange(50, dtype=np.int32), topk=7)
    want = np.argsort(-(algo.P @ algo.Q.T), axis=1)[:, :7]
    assert (topks == want).mean() > 0.99 and scores.shape == (50, 7)
    keys, topks, scores = par.ParBPRMF(algo).topk_recommendation(np.arange(50, dtype=np.int32), topk=7, pool=np.arange(0, 80, 2, dtype=np.int32))
    assert np.all(topks % 2 == 0)
    topks, scores = par.ParALS(algo).most_similar(np.arange(10, dtype=np.int32), topk=5, group="item")
    assert np.all(topks != np.arange(10)[:, None])
    with pytest.raises(RuntimeError):
        par.ParALS(algo).topk_recommendation(np.arange(3, dtype=np.int32), pool=[])


def test_full_size_validation_sweep_properties():
    """BASELINE-sized consumer: 4096 users x 27,278 items, d=128, k=100: sortedness, no duplicates, and a
    float64 check of sampled rows (every returned score is the row's true score, and nothing outside the
    list beats the last entry by more than rounding)."""
    from buffalo_amd import parallel as par
    rng = np.random.default_rng(0)
    P = rng.normal(scale=0.1, size=(4096, 128)).astype(np.float32)
    Q = rng.normal(scale=0.1, size=(27278, 128)).astype(np.float32)
    idx = np.arange(4096, dtype=np.int32)
    keys, scores = tc.run(par.dot_topn, idx, P, Q, tc.NO_BIAS, tc.EMPTY_POOL, 100)
    assert np.all(keys >= 0) and np.all(np.diff(scores, axis=1) <= 0)
    assert all(len(set(r)) == 100 for r in keys[::64])
    for q in range(0, 4096, 512):
        full = P[q].astype(np.float64) @ Q.T.astype(np.float64)
        assert np.abs(full[keys[q]] - scores[q]).max() < 1e-5
        rest = np.delete(full, keys[q])
        assert rest.max() <= scores[q, -1] + 1e-5
