"""Pins for the CPU oracle: the reference has no golden vectors for its training arithmetic
(SURVEY.md section 4), so the C++ restatement is checked against independent numpy transliterations
(tests/ref_numpy.py), analytic micro-cases and published known-answer vectors."""
import os

import numpy as np
import pytest

from conftest import als_opt, bpr_opt, tiny_csr, warp_opt
import ref_numpy as rn


def test_philox_known_answers(oracle):
    for ctr, key, want in rn.PHILOX_KAT:
        assert tuple(rn.philox4x32_10(ctr, key)) == want
        assert tuple(oracle.philox4x32_10(ctr, key)) == want
    for args in [(7, 0, 0, 0, 0, 0), (777, 1, 20000262, 3, 9, 41), (1, 0, 2 ** 33 + 5, 0, 1, 2)]:
        assert oracle.counter_draw(*args) == rn.counter_draw(*args)


def test_exp_table_and_index(oracle):
    t = oracle.OracleBPRMF().exp_table()
    want = rn.exp_table()
    np.testing.assert_allclose(t, want, rtol=3e-7)  # expf implementations differ by <= 1 ulp
    # analytic: entry i is sigmoid(-x_i), x_i = (2i/1000 - 1) * 6
    for i in (0, 500, 999):
        x = (i / 1000.0 * 2 - 1) * 6
        assert abs(float(t[i]) - 1.0 / (np.exp(x) + 1.0)) < 1e-6
    assert abs(float(t[500]) - 0.5) < 1e-7
    assert 1000 // 6 // 2 == 83 and int((6.0 + 6) * 83) == 996  # Q-2: max index reachable


def _bpr(oracle, opt_file, csr, P, Q, Qb, **kw):
    o = oracle.OracleBPRMF()
    assert o.init(opt_file(bpr_opt(**kw)))
    o.initialize_model(P, Q, Qb, csr.nnz)
    o.set_cumulative_table(np.zeros(csr.num_items, dtype=np.int64), csr.num_items)
    return o


def test_bpr_single_triple_analytic(oracle, opt_file):
    """1 user, 2 items, d=4: the negative is forced (verify_neg rejects the only positive)."""
    from buffalo_amd.synth import CSR
    csr = CSR(1, 2, [1], [0], [1.0])
    P = np.array([[0.5, -0.25, 0.125, 1.0]], dtype=np.float32)
    Q = np.array([[0.2, 0.1, -0.3, 0.4], [-0.1, 0.3, 0.2, 0.05]], dtype=np.float32)
    Qb = np.array([[0.01], [-0.02]], dtype=np.float32)
    kw = dict(d=4, lr=0.1, min_lr=0.1, num_iters=1, random_seed=5)
    P0, Q0, Qb0 = P.copy(), Q.copy(), Qb.copy()
    o = _bpr(oracle, opt_file, csr, P, Q, Qb, **kw)
    o.set_modes(inline=True)
    o.launch_workers()
    o.add_jobs(0, 1, csr.indptr, csr.keys)
    o.join()
    opt = bpr_opt(**kw)
    rn.bpr_sgd_step(P0, Q0, Qb0, 0, 0, 1, 0.1, opt, rn.exp_table())
    np.testing.assert_allclose(P, P0, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(Q, Q0, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(Qb, Qb0, rtol=1e-6, atol=1e-8)
    # hand formula for the user row makes Q-1 explicit: P uses the UPDATED item rows
    x = float(np.dot([0.5, -0.25, 0.125, 1.0], np.array([0.2, 0.1, -0.3, 0.4]) - [-0.1, 0.3, 0.2, 0.05])
              + 0.03)
    logit = float(rn.exp_table()[int((x + 6) * 83)])
    assert abs(logit - 1 / (1 + np.exp(x))) < 2e-2  # LUT (index step 1/83, table step 12/1000) is only ~1e-2 accurate


@pytest.mark.parametrize("optimizer", ["sgd", "adagrad", "adam"])
def test_bpr_epoch_matches_transliteration(oracle, opt_file, optimizer):
    """Replay the oracle's own (u,pos,neg) trace through the numpy transliteration."""
    csr = tiny_csr(U=10, I=14, seed=11)
    d = 6
    rng = np.random.default_rng(1)
    P = rng.normal(scale=0.3, size=(csr.num_users, d)).astype(np.float32)
    Q = rng.normal(scale=0.3, size=(csr.num_items, d)).astype(np.float32)
    Qb = rng.normal(scale=0.1, size=(csr.num_items, 1)).astype(np.float32)
    kw = dict(d=d, lr=0.05, min_lr=0.05, num_iters=2, random_seed=7, optimizer=optimizer,
              num_negative_samples=2, per_coordinate_normalize=(optimizer == "adam"))
    P0, Q0, Qb0 = P.copy(), Q.copy(), Qb.copy()
    o = _bpr(oracle, opt_file, csr, P, Q, Qb, **kw)
    o.set_modes(inline=True)
    o.trace(True)
    o.launch_workers()
    opt = bpr_opt(**kw)
    table = rn.exp_table()
    gP, gQ, gQb = np.zeros_like(P0), np.zeros_like(Q0), np.zeros(csr.num_items, dtype=np.float32)
    st = {k: np.zeros_like(v) for k, v in (("mP", P0), ("vP", P0), ("mQ", Q0), ("vQ", Q0))}
    mb, vb = np.zeros((csr.num_items, 1), np.float32), np.zeros((csr.num_items, 1), np.float32)
    seen_total = 0
    for it in range(2):
        o.add_jobs(0, csr.num_users, csr.indptr, csr.keys)
        tr = o.get_trace()[seen_total:]
        seen_total += len(tr)
        assert len(tr) == csr.nnz * 2
        cntP, cntQ = np.zeros(csr.num_users, int), np.zeros(csr.num_items, int)
        for k, (u, pos, neg) in enumerate(tr):
            if optimizer == "sgd":
                rn.bpr_sgd_step(P0, Q0, Qb0, u, pos, neg, 0.05, opt, table)
            else:
                rn.bpr_accumulate_step(P0, Q0, Qb0, gP, gQ, gQb, u, pos, neg, opt, table)
                cntQ[neg] += 1            # bpr.cc:139-143: per negative
                if k % 2 == 1:            # bpr.cc:175-181: per positive (after its negatives)
                    cntP[u] += 1
                    cntQ[pos] += 1
        o.update_parameters()
        if optimizer != "sgd":
            g2 = gQb.reshape(-1, 1)
            if opt["per_coordinate_normalize"]:  # Q-9: gradQb is divided together with gradQ
                nz = cntQ > 0
                g2[nz, 0] = g2[nz, 0] / cntQ[nz].astype(np.float32)
            rn.update_parameters(P0, gP, st["mP"], st["vP"], cntP, opt["reg_u"], opt, it)
            rn.update_parameters(Q0, gQ, st["mQ"], st["vQ"], cntQ, opt["reg_i"], opt, it)
            o2 = dict(opt, per_coordinate_normalize=False)
            rn.update_parameters(Qb0, g2, mb, vb, None, opt["reg_b"], o2, it)
    o.join()
    np.testing.assert_allclose(P, P0, rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(Q, Q0, rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(Qb, Qb0, rtol=2e-5, atol=2e-7)
    if optimizer != "sgd":  # Q-6: grad buffers keep the transformed step (never re-zeroed)
        np.testing.assert_allclose(o.state("gradP").reshape(P.shape), gP, rtol=2e-5, atol=2e-7)
        assert np.abs(o.state("gradQ")).max() > 0


def test_bpr_apply_triples_replays_any_schedule(oracle, opt_file):
    """`apply_triples` (the SGD step of bpr.cc:119-171 over an explicit triple list) is what the GPU schedules are
    replayed through: (a) fed the oracle's own trace it reproduces the oracle's epoch bit for bit; (b) fed the same
    triples in another order (the item-major walk's) it lands on a different model -- the one the numpy
    transliteration reaches in that order."""
    from helpers import item_major_order as _item_major_order
    csr = tiny_csr(U=60, I=40, density=0.15, seed=4)
    d, nn = 12, 2
    rng = np.random.default_rng(2)
    P = rng.normal(scale=0.3, size=(csr.num_users, d)).astype(np.float32)
    Q = rng.normal(scale=0.3, size=(csr.num_items, d)).astype(np.float32)
    Qb = rng.normal(scale=0.1, size=(csr.num_items, 1)).astype(np.float32)
    kw = dict(d=d, lr=0.05, min_lr=0.05, num_iters=1, random_seed=9, num_negative_samples=nn)
    P0, Q0, Qb0 = P.copy(), Q.copy(), Qb.copy()
    o = _bpr(oracle, opt_file, csr, P, Q, Qb, **kw)
    o.set_modes(sampler="counter", pos_order="csr", inline=True)
    o.trace(True)
    o.launch_workers()
    o.add_jobs(0, csr.num_users, csr.indptr, csr.keys)
    tr = o.get_trace()
    o.join()
    assert len(tr) == csr.nnz * nn

    def replay(order):
        Pr, Qr, Qbr = P0.copy(), Q0.copy(), Qb0.copy()
        r = _bpr(oracle, opt_file, csr, Pr, Qr, Qbr, **kw)
        t = tr[order]
        r.apply_triples(np.ascontiguousarray(t[:, 0]), np.ascontiguousarray(t[:, 1]), np.ascontiguousarray(t[:, 2]), 0.05)
        return Pr, Qr, Qbr

    Pa, Qa, Qba = replay(np.arange(len(tr)))
    np.testing.assert_array_equal(Pa, P)
    np.testing.assert_array_equal(Qa, Q)
    np.testing.assert_array_equal(Qba, Qb)
    order = _item_major_order(csr, 8, 3, nn)
    Pb, Qb_im, Qbb = replay(order)
    assert np.abs(Pb - P).max() > 1e-3                      # another order, another model
    Pn, Qn, Qbn = P0.copy(), Q0.copy(), Qb0.copy()
    opt, table = bpr_opt(**kw), rn.exp_table()
    for u, pos, neg in tr[order]:
        rn.bpr_sgd_step(Pn, Qn, Qbn, u, pos, neg, 0.05, opt, table)
    np.testing.assert_allclose(Pb, Pn, rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(Qb_im, Qn, rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(Qbb, Qbn, rtol=2e-5, atol=2e-7)


def test_bpr_sampling_quirks(oracle, opt_file):
    """Q-3 (unordered_set order), Q-4 (lower_bound on int64 cumulative counts), verify_neg."""
    csr = tiny_csr(U=8, I=20, seed=5)
    d = 4
    P, Q, Qb = [np.zeros((n, c), np.float32) for n, c in ((8, d), (20, d), (20, 1))]
    o = _bpr(oracle, opt_file, csr, P, Q, Qb, d=d, random_seed=3, sampling_power=1.0, lr=0.0, min_lr=0.0)
    cnt = np.bincount(csr.keys, minlength=20).astype(np.int64)
    cum = np.cumsum(cnt)
    o.set_cumulative_table(cum, 20)
    o.set_modes(inline=True)
    o.trace(True)
    o.launch_workers()
    for _ in range(30):
        o.add_jobs(0, 8, csr.indptr, csr.keys)
    tr = o.get_trace()
    for u in range(8):
        keys, _ = csr.row(u)
        m = tr[:, 0] == u
        assert set(tr[m, 1]) == set(keys)               # every positive visited
        assert not (set(tr[m, 2]) & set(keys))          # verify_neg
        first = tr[m, 1][:len(keys)]
        assert sorted(first) == sorted(keys)
    # popularity sampling: an item with zero count whose predecessor has cum==r can still be drawn
    # (lower_bound quirk), but items are overwhelmingly drawn ~ counts
    neg_hist = np.bincount(tr[:, 2], minlength=20)
    assert neg_hist[np.argmax(cnt)] > neg_hist[np.argmin(cnt)]


def test_threaded_equals_inline_with_one_worker(oracle, opt_file):
    """Q-8: with min_lr == lr the reference path is deterministic for num_workers == 1."""
    csr = tiny_csr(U=30, I=40, seed=9)
    d = 8
    outs = []
    for inline in (False, True):
        rng = np.random.default_rng(2)
        P = rng.normal(scale=0.2, size=(30, d)).astype(np.float32)
        Q = rng.normal(scale=0.2, size=(40, d)).astype(np.float32)
        Qb = np.zeros((40, 1), np.float32)
        o = _bpr(oracle, opt_file, csr, P, Q, Qb, d=d, lr=0.03, min_lr=0.03, num_iters=3, random_seed=7)
        o.set_modes(inline=inline)
        o.launch_workers()
        for _ in range(3):
            o.add_jobs(0, 30, csr.indptr, csr.keys)
            o.update_parameters()
            o.wait_until_done()
        o.join()
        outs.append((P, Q, Qb))
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


def test_inline_lr_decay_per_call(oracle, opt_file):
    """Deterministic idealisation of Q-8: lr(call) = max(min_lr, lr - (lr-min_lr)*processed/total),
    processed counting job.size = sum(1 + n_pos)."""
    from buffalo_amd.synth import CSR
    csr = CSR(2, 3, [1, 2], [0, 1], [1.0, 1.0])
    P = np.array([[1.0], [1.0]], np.float32)
    Q = np.zeros((3, 1), np.float32)
    Qb = np.zeros((3, 1), np.float32)
    kw = dict(d=1, lr=0.5, min_lr=0.1, num_iters=2, reg_u=1.0, reg_i=0, reg_j=0, reg_b=0,
              update_i=False, update_j=False, use_bias=False)
    o = _bpr(oracle, opt_file, csr, P, Q, Qb, **kw)
    o.set_modes(inline=True)
    o.launch_workers()
    o.add_jobs(0, 2, csr.indptr, csr.keys)   # lr = 0.5: P *= (1 - 0.5)
    np.testing.assert_allclose(P, [[0.5], [0.5]], rtol=1e-6)
    o.add_jobs(0, 2, csr.indptr, csr.keys)   # progress = 4 / (2*2) = 1 -> lr = 0.1
    np.testing.assert_allclose(P, [[0.45], [0.45]], rtol=1e-6)


def test_warp_epoch_matches_transliteration(oracle, opt_file):
    csr = tiny_csr(U=9, I=25, seed=4)
    d = 5
    rng = np.random.default_rng(8)
    P = rng.normal(scale=0.8, size=(9, d)).astype(np.float32)
    Q = rng.normal(scale=0.8, size=(25, d)).astype(np.float32)
    Qb = np.zeros((25, 1), np.float32)
    kw = dict(d=d, random_seed=13, max_trials=12, threshold=0.6, reg_u=0.01, reg_i=0.02, reg_j=0.03,
              num_iters=2, lr=0.05)
    o = oracle.OracleWARP()
    assert o.init(opt_file(warp_opt(**kw)))
    P0, Q0 = P.copy(), Q.copy()
    o.initialize_model(P, Q, Qb, csr.nnz)
    o.set_modes(sampler="counter", pos_order="csr", inline=True)
    o.launch_workers()
    opt = warp_opt(**kw)
    gP, gQ = np.zeros_like(P0), np.zeros_like(Q0)
    vP, vQ = np.zeros_like(P0), np.zeros_like(Q0)
    scored = accepted = 0
    for epoch in range(2):
        o.add_jobs(0, 9, csr.indptr, csr.keys)
        for u in range(9):
            keys, _ = csr.row(u)
            beg = 0 if u == 0 else int(csr.indptr[u - 1])
            seen = set(int(k) for k in keys)
            for k, pos in enumerate(keys):
                draw = lambda a, idx=beg + k: (rn.counter_draw(13, 1, idx, 0, epoch, a)[0] * 25) >> 32
                ok, _, trial, ns = rn.warp_positive(P0, Q0, gP, gQ, u, int(pos), seen, draw, opt)
                scored += ns
                accepted += int(ok)
        o.update_parameters()
        rn.update_parameters(P0, gP, None, vP, None, opt["reg_u"], opt, epoch)
        rn.update_parameters(Q0, gQ, None, vQ, None, opt["reg_i"], opt, epoch)
        rn.unit_ball_project(Q0)
        rn.unit_ball_project(P0)
    st = o.stats()
    assert st["scored_negatives"] == scored and st["updates"] == accepted
    assert 0 < accepted < csr.nnz * 2  # both the accept and the give-up branch were exercised
    np.testing.assert_allclose(P, P0, rtol=3e-5, atol=3e-7)
    np.testing.assert_allclose(Q, Q0, rtol=3e-5, atol=3e-7)


def test_warp_trial_counting_q10(oracle, opt_file):
    """k-th counted try accepted => trial == 2k; skipped when trial >= max_trials."""
    P = np.array([[1.0, 0.0]], np.float32)
    Q = np.array([[5.0, 0], [0.0, 0], [0.0, 0], [0.0, 0]], np.float32)  # ui - uj = 5 > threshold: never violates
    gP, gQ = np.zeros_like(P), np.zeros_like(Q)
    opt = warp_opt(max_trials=7, threshold=1.0)
    ok, _, trial, scored = rn.warp_positive(P, Q, gP, gQ, 0, 0, {0}, lambda a: 1 + a % 3, opt)
    assert not ok and scored == 4 and trial == 9          # 1 -> 3 -> 5 -> 7 -> 9 (> max_trials)
    Q[2, 0] = 4.5                                         # violator at the 2nd counted try
    ok, neg, trial, scored = rn.warp_positive(P, Q, gP, gQ, 0, 0, {0}, lambda a: 1 + a % 3, opt)
    assert ok and neg == 2 and trial == 4 and scored == 2
    Phi = np.log(max(1, (4 - 1 - 1) // 4))
    assert Phi == 0.0 and np.all(gQ == 0)                 # Phi = log(max(1, 0)) = 0: zero update


@pytest.mark.parametrize("optimizer", ["llt", "ldlt", "manual_cg"])
@pytest.mark.parametrize("axis", [0, 1])
def test_als_dense_row_solve(oracle, opt_file, optimizer, axis):
    csr = tiny_csr(U=11, I=13, seed=6, counts=True)
    mat = csr if axis == 0 else csr.transpose()
    d = 7
    rng = np.random.default_rng(3)
    P = np.abs(rng.normal(scale=0.3, size=(11, d))).astype(np.float32)
    Q = np.abs(rng.normal(scale=0.3, size=(13, d))).astype(np.float32)
    kw = dict(d=d, optimizer=optimizer, alpha=4.0, reg_u=0.2, reg_i=0.3, adaptive_reg=(axis == 1))
    o = oracle.OracleALS()
    assert o.init(opt_file(als_opt(**kw)))
    P0, Q0 = P.copy(), Q.copy()
    o.initialize_model(P, Q)
    o.precompute(axis)
    X0, Y0 = (P0, Q0) if axis == 0 else (Q0, P0)
    FF = (Y0.astype(np.float64).T @ Y0.astype(np.float64)).astype(np.float32)
    np.testing.assert_allclose(o.get_ff(d), FF, rtol=1e-5, atol=1e-6)
    nume, deno = o.partial_update(0, mat.num_users, mat.indptr, mat.keys, mat.vals, axis)
    reg = kw["reg_u"] if axis == 0 else kw["reg_i"]
    X = P if axis == 0 else Q
    wn = wd = 0.0
    for u in range(mat.num_users):
        keys, vals = mat.row(u)
        A, y = rn.als_normal_equations(X0, Y0, FF, u, keys, vals, 4.0, reg, kw["adaptive_reg"])
        n_, d_ = rn.als_loss_terms(X0, Y0, FF, u, keys, vals, 4.0, reg, kw["adaptive_reg"], axis)
        wn, wd = wn + n_, wd + d_
        if optimizer == "manual_cg":
            want = rn.manual_cg(X0[u], A, y)
            assert np.abs(X[u] - want).max() <= 1e-4 * np.abs(want).max()  # 3 fp32 CG steps
        else:
            want = np.linalg.solve(A, y)
            np.testing.assert_allclose(X[u], want, rtol=5e-4, atol=5e-6)
    assert abs(nume - wn) <= 1e-4 * max(1.0, abs(wn)) and abs(deno - wd) <= 1e-4 * max(1.0, abs(wd))


def test_als_empty_rows_untouched_q16(oracle, opt_file):
    from buffalo_amd.synth import CSR
    csr = CSR(3, 4, [2, 2, 3], [0, 3, 1], [1, 2, 1])
    P = np.full((3, 4), 0.25, np.float32)
    Q = np.abs(np.random.default_rng(0).normal(size=(4, 4))).astype(np.float32)
    o = oracle.OracleALS()
    assert o.init(opt_file(als_opt(d=4, optimizer="llt")))
    o.initialize_model(P, Q)
    o.precompute(0)
    o.partial_update(0, 3, csr.indptr, csr.keys, csr.vals, 0)
    assert np.all(P[1] == 0.25) and not np.allclose(P[0], 0.25)


@pytest.mark.parametrize("d,block", [(12, 5), (128, 32)])
def test_ialspp_matches_transliteration(oracle, opt_file, d, block):
    csr = tiny_csr(U=6, I=9, seed=8, counts=True)
    rng = np.random.default_rng(5)
    P = np.abs(rng.normal(scale=0.2, size=(6, d))).astype(np.float32)
    Q = np.abs(rng.normal(scale=0.2, size=(9, d))).astype(np.float32)
    o = oracle.OracleALS()
    assert o.init(opt_file(als_opt(d=d, optimizer="ialspp" if d < 128 else "manual_cg",  # Q-13
                                   block_size=block, alpha=3.0, reg_u=0.15)))
    P0, Q0 = P.copy(), Q.copy()
    o.initialize_model(P, Q)
    o.precompute(0)
    FF = o.get_ff(d)
    o.partial_update(0, 6, csr.indptr, csr.keys, csr.vals, 0)
    for u in range(6):
        keys, vals = csr.row(u)
        want = rn.ialspp_row(P0, Q0, FF, u, keys, vals, 3.0, 0.15, block)
        assert np.abs(P[u] - want).max() <= 3e-4 * np.abs(want).max()  # 3 fp32 CG steps per block
    assert not np.allclose(P, P0)


# ------------------------------------------------------------------------------------------------
# parallel::dot_topn / quickselect restatement -- pinned on the reference's
# own parallel tests (tests/parallel/test_base.py:38-101, numpy argsort as the known answer) and on
# exact-arithmetic cases for the admission / tie rules of _core.hpp:37-67,115-137
# ------------------------------------------------------------------------------------------------
@pytest.mark.skipif(not os.path.isdir("/root/reference/tests/parallel"), reason="/root/reference is not here")
def test_topn_the_reference_s_own_parallel_tests_pass_on_the_oracle():
    """tests/parallel/test_base.py of the reference, UNMODIFIED (test00, 01, 03, 04; 02 is a thread-scaling timing test), with
    `buffalo.parallel._core.dot_topn` bound to the oracle: buffalo/parallel/base.py is imported from /root/reference as it is
    (tests/golden/run_reference_tests.py).  The three tests below restate the same cases for machines without the reference."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "golden", "run_reference_tests.py"), "parallel"], capture_output=True, text=True)
    assert r.returncode == 0 and "ran 4, failures 0, errors 0" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_topn_reference_test01_most_similar(oracle):
    import topk_cases as tc
    Q = tc.unit_factors(128, 5, seed=1)
    indexes = np.array([0, 1, 2, 3, 4], dtype=np.int32)
    keys, scores = tc.run(oracle.dot_topn, indexes, Q, Q, tc.NO_BIAS, tc.EMPTY_POOL, 10)
    wk, ws = tc.numpy_most_similar(indexes, Q, 10)
    assert np.array_equal(keys, wk) and np.allclose(scores, ws, atol=1e-7)


def test_topn_reference_test03_pool(oracle):
    import topk_cases as tc
    Q = tc.unit_factors(128, 5, seed=2)
    keys, scores = tc.run(oracle.dot_topn, np.arange(5, dtype=np.int32), Q, Q, tc.NO_BIAS, np.array([5, 6, 7], np.int32), 10)
    assert set(keys.reshape(-1)) == {5, 6, 7, -1}
    assert np.all(keys[:, 3:] == -1) and np.all(scores[:, 3:] == 0.0)       # slots >= correct_k: (-1, 0.0)
    assert np.all(np.diff(scores[:, :3], axis=1) <= 0)


def test_topn_reference_test04_topk(oracle):
    import topk_cases as tc
    P, Q = tc.unit_factors(512, 5, seed=3), tc.unit_factors(128, 5, seed=4)
    idx = np.array([312, 313, 314, 315, 316], dtype=np.int32)
    keys, scores = tc.run(oracle.dot_topn, idx, P, Q, tc.NO_BIAS, tc.EMPTY_POOL, 10)
    wk, ws = tc.numpy_topk(idx, P, Q, 10)
    assert np.array_equal(keys, wk) and np.allclose(scores, ws, atol=1e-7)


@pytest.mark.parametrize("same,bias,pool,k", [(False, False, [], 7), (True, False, [], 7), (False, True, [], 40),
                                              (True, True, [1, 2, 3, 5, 8, 13, 21, 34], 5), (False, False, [0, 4], 6)])
def test_topn_admission_and_tie_rules_exact(oracle, same, bias, pool, k):
    """Integer factors => exact scores with many ties, zeros and negatives: non-positive scores are never
    admitted (FLT_MIN start value), unfilled slots read (-1, FLT_MIN), equal scores are listed by DESCENDING
    index, and boundary ties follow the admit-early / evict-oldest rule spelled out in topk_cases.spec_dot_topn."""
    import topk_cases as tc
    Q = tc.integer_factors(37, 6, seed=5)
    P = Q if same else tc.integer_factors(9, 6, seed=6)
    Qb = tc.integer_factors(37, 1, seed=7, lo=-1, hi=2) if bias else tc.NO_BIAS
    idx = np.arange(P.shape[0] if not same else 9, dtype=np.int32)
    keys, scores = tc.run(oracle.dot_topn, idx, P, Q, Qb, np.array(pool, np.int32), k)
    wk, ws = tc.spec_dot_topn(idx, P, Q, Qb, pool, k, same)
    assert np.array_equal(keys, wk)
    assert np.array_equal(scores, ws)
    assert (keys == -1).any() or k <= 7        # the cases do exercise unfilled slots


def test_quickselect_matches_numpy_on_distinct_scores(oracle):
    rng = np.random.default_rng(11)
    scores = rng.permutation(64 * 200).reshape(64, 200).astype(np.float32)   # all distinct
    for k, srt in ((1, True), (10, True), (200, True), (25, False)):
        res = np.empty((64, k), np.int32)
        oracle.quickselect(scores, res, srt)
        want = np.argsort(-scores, axis=1)[:, :k]
        if srt:
            assert np.array_equal(res, want)
        else:
            assert np.array_equal(np.sort(res, axis=1), np.sort(want, axis=1))


def test_coo_to_csr_matches_scipy_and_is_stable(oracle):
    """fileio.hpp:263-420: stable (row, col) sort keeping duplicates, END-offset indptr."""
    import scipy.sparse as sp
    rng = np.random.default_rng(4)
    n, R, Cc = 5000, 70, 40
    r = rng.integers(0, R, n).astype(np.int32)
    r[r == 13] = 14                                   # an empty row in the middle; rows 0.. may be empty at the ends too
    c = rng.integers(0, Cc, n).astype(np.int32)
    v = np.arange(n, dtype=np.float32)               # distinct values make the record order visible
    g = oracle.coo_to_csr(r, c, v, R + 3, Cc)         # three trailing empty rows
    order = np.lexsort((np.arange(n), c, r))          # stable: input order breaks (row, col) ties
    assert np.array_equal(g["key"], c[order]) and np.array_equal(g["val"], v[order])
    want = np.cumsum(np.bincount(r, minlength=R + 3))
    assert np.array_equal(g["indptr"], want) and g["indptr"][13] == g["indptr"][12] and g["indptr"][-1] == n
    M = sp.coo_matrix((np.ones(n, np.float32), (r, c)), shape=(R + 3, Cc)).tocsr()   # scipy sums duplicates: compare the pattern
    M.sort_indices()
    assert np.array_equal(np.unique(np.stack([np.repeat(np.arange(R + 3), np.diff(np.concatenate([[0], g["indptr"]]))), g["key"]]), axis=1),
                          np.stack([np.repeat(np.arange(R + 3), np.diff(M.indptr)), M.indices]))


# ------------------------------------------------------------------------------------------------
# CCFR restatement against a float64 evaluation of the normal equations of
# cfr.cc:92-313 written directly from the paper's objective (user / item / context rows, biases, losses)
# ------------------------------------------------------------------------------------------------
def test_cfr_rows_biases_and_losses_match_numpy(oracle, opt_file):
    Uu, Ii, d = 14, 9, 6
    csr = tiny_csr(U=Uu, I=Ii, density=0.4, seed=1, counts=True)
    t = csr.transpose()
    ctx = tiny_csr(U=Ii, I=Ii, density=0.35, seed=2, counts=True)
    alpha, l, ru, ri, rc = 3.0, 0.6, 0.1, 0.2, 0.3
    opt = {"d": d, "num_workers": 2, "num_cg_max_iters": 3, "alpha": alpha, "l": l, "eps": 1e-10, "reg_u": ru, "reg_i": ri, "reg_c": rc,
           "compute_loss": True, "optimizer": "llt", "cg_tolerance": 1e-10}
    rng = np.random.default_rng(0)
    f = lambda r, c: rng.normal(scale=0.3, size=(r, c)).astype(np.float32)     # noqa: E731
    U, I, Cx, Ib, Cb = f(Uu, d), f(Ii, d), f(Ii, d), f(Ii, 1), f(Ii, 1)
    o = oracle.OracleCFR()
    assert o.init(opt_file(opt))
    for F, n in ((U, "user"), (I, "item"), (Cx, "context"), (Ib, "item_bias"), (Cb, "context_bias")):
        o.set_embedding(F, n)
    f64 = lambda a: a.astype(np.float64)     # noqa: E731
    # ---- users (cfr.cc:92-146)
    I0 = f64(I)
    FF = I0.T @ I0
    want_U = f64(U).copy()
    for x in range(Uu):
        k, v = csr.row(x)
        Fs = I0[k]
        A = (FF + Fs.T @ (Fs * (alpha * v)[:, None])) * l + ru * np.eye(d)
        want_U[x] = np.linalg.solve(A, ((1 + alpha * v) @ Fs) * l)
    o.precompute("item")
    loss_u = o.partial_update_user(0, Uu, csr.indptr, csr.keys, csr.vals)
    np.testing.assert_allclose(U, want_U, rtol=2e-4, atol=2e-5)
    assert abs(loss_u - ru * (f64(U) ** 2).sum()) < 1e-4 * max(1.0, loss_u)       # the UPDATED rows
    # ---- items (cfr.cc:148-255)
    U0, C0, Ib0, Cb0, Iold = f64(U), f64(Cx), f64(Ib).ravel(), f64(Cb).ravel(), f64(I).copy()
    FFu = U0.T @ U0
    want_I, want_Ib, want_loss = Iold.copy(), Ib0.copy(), 0.0
    for x in range(Ii):
        ku, vu = t.row(x)
        kc, vc = ctx.row(x)
        Fu, Fc = U0[ku], C0[kc]
        dots = Fu @ Iold[x]
        want_loss += l * (Iold[x] @ FFu @ Iold[x] + (-dots ** 2 + (1 + alpha * vu) * (dots - 1) ** 2).sum())
        want_loss += ((vc - Fc @ Iold[x] - Ib0[x] - Cb0[kc]) ** 2).sum() + ri * Iold[x] @ Iold[x]
        A = (FFu + Fu.T @ (Fu * (alpha * vu)[:, None])) * l + Fc.T @ Fc + ri * np.eye(d)
        y = ((1 + alpha * vu) @ Fu) * l + (vc - Ib0[x] - Cb0[kc]) @ Fc
        want_I[x] = np.linalg.solve(A, y)
        want_Ib[x] = (vc - Fc @ want_I[x] - Cb0[kc]).sum() / (len(kc) + 1e-10)
    o.precompute("user")
    loss_i = o.partial_update_item(0, Ii, t.indptr, t.keys, t.vals, ctx.indptr, ctx.keys, ctx.vals)
    np.testing.assert_allclose(I, want_I, rtol=5e-4, atol=5e-5)
    np.testing.assert_allclose(Ib.ravel(), want_Ib, rtol=5e-4, atol=5e-5)
    assert abs(loss_i - want_loss) < 2e-4 * max(1.0, abs(want_loss))
    # ---- contexts (cfr.cc:257-313)
    I1, Ib1, Cold = f64(I), f64(Ib).ravel(), f64(Cx).copy()
    want_C, want_Cb = Cold.copy(), Cb0.copy()
    for x in range(Ii):
        k, v = ctx.row(x)
        Fs = I1[k]
        want_C[x] = np.linalg.solve(Fs.T @ Fs + rc * np.eye(d), (v - Cb0[x] - Ib1[k]) @ Fs)
        want_Cb[x] = (v - Fs @ want_C[x] - Ib1[k]).sum() / (len(k) + 1e-10)
    loss_c = o.partial_update_context(0, Ii, ctx.indptr, ctx.keys, ctx.vals)
    np.testing.assert_allclose(Cx, want_C, rtol=5e-4, atol=5e-5)
    np.testing.assert_allclose(Cb.ravel(), want_Cb, rtol=5e-4, atol=5e-5)
    assert abs(loss_c - rc * (Cold ** 2).sum()) < 1e-4 * max(1.0, loss_c)          # the rows BEFORE the update


def test_eals_descends_the_float64_objective_and_keeps_its_cache(oracle, opt_file):
    """CEALS restatement: every half-epoch is an exact coordinate minimisation of
        sum_obs (1 + alpha v)(v - p.q)^2 + sum_unobs c_i (p.q)^2 + reg_u |P|^2 + reg_i |Q|^2
    (eals.cc:122-125), so the loss `estimate_loss` derives from the prediction cache must equal that objective
    recomputed from P, Q in float64, never increase, and the index maps must link the two orientations."""
    csr = tiny_csr(U=25, I=18, density=0.3, seed=1, counts=True)
    t = csr.transpose()
    d, alpha, ru, ri = 6, 2.0, 0.1, 0.2
    rng = np.random.default_rng(0)
    P = rng.normal(scale=0.3, size=(25, d)).astype(np.float32)
    Q = rng.normal(scale=0.3, size=(18, d)).astype(np.float32)
    pop = np.bincount(csr.keys, minlength=18).astype(np.float64) ** 0.5
    Cw = (0.5 * pop / pop.sum()).astype(np.float32)
    o = oracle.OracleEALS()
    assert o.init(opt_file({"d": d, "num_workers": 2, "alpha": alpha, "reg_u": ru, "reg_i": ri}))
    o.initialize_model(P, Q, Cw)
    assert not o.update(csr.indptr, csr.keys, csr.vals, 0)                 # eals.cc:106-114: no cache, no update
    assert o.estimate_loss(csr.nnz, csr.indptr, csr.keys, csr.vals, 0) == (0.0, 0.0)
    o.precompute_cache(csr.nnz, csr.indptr, csr.keys, 0)
    o.precompute_cache(csr.nnz, t.indptr, t.keys, 1)
    rows = np.repeat(np.arange(25), np.diff(np.concatenate([[0], csr.indptr])))
    trows = np.repeat(np.arange(18), np.diff(np.concatenate([[0], t.indptr])))
    pos = {(int(u), int(i)): r for r, (i, u) in enumerate(zip(trows, t.keys))}
    _, u2i = o.caches(0, csr.nnz)
    assert all(pos[(int(rows[k]), int(csr.keys[k]))] == u2i[k] for k in range(csr.nnz))

    def objective():
        P64, Q64 = P.astype(np.float64), Q.astype(np.float64)
        S = P64 @ Q64.T
        W = np.tile(Cw.astype(np.float64), (25, 1))
        R = np.zeros_like(S)
        R[rows, csr.keys] = csr.vals
        W[rows, csr.keys] = 1 + alpha * csr.vals
        return (W * (R - S) ** 2).sum() + ru * (P64 ** 2).sum() + ri * (Q64 ** 2).sum()
    prev = objective()
    assert abs(o.estimate_loss(csr.nnz, csr.indptr, csr.keys, csr.vals, 0)[1] - prev) < 1e-4 * prev
    for _ in range(4):
        for axis, m in ((0, csr), (1, t)):
            assert o.update(m.indptr, m.keys, m.vals, axis)
            cur = objective()
            assert cur <= prev * (1 + 1e-6)
            prev = cur
    rmse, loss = o.estimate_loss(csr.nnz, t.indptr, t.keys, t.vals, 1)
    assert abs(loss - prev) < 1e-4 * prev
    vh, _ = o.caches(0, csr.nnz)
    assert np.abs(vh - (P[rows] * Q[csr.keys]).sum(1)).max() < 1e-5
    assert abs(rmse - np.sqrt(((csr.vals - vh) ** 2).mean())) < 1e-5


def test_ialspp_f64_fast_equals_loop_version():
    """tests/ref_numpy.py: the matrix-product form of the float64 iALS++ row recurrence (used at config #3 size) against the
    per-entry loop form the oracle is pinned on."""
    import ref_numpy as R
    rng = np.random.default_rng(3)
    D, n = 96, 57
    P = rng.normal(scale=0.2, size=(3, D)).astype(np.float32)
    Q = rng.normal(scale=0.2, size=(40, D)).astype(np.float32)
    FF = (Q.astype(np.float64).T @ Q.astype(np.float64)).astype(np.float32)
    keys = rng.integers(0, 40, n)
    vals = (1 + rng.poisson(1.0, n)).astype(np.float32)
    for bs in (32, 7):
        a = R.ialspp_row_f64(P, Q, FF, 1, keys, vals, 4.0, 0.3, bs)
        b = R.ialspp_row_f64_fast(P[1], Q[keys], FF, vals, 4.0, 0.3, bs)
        assert np.abs(a - b).max() < 1e-12 * max(1.0, np.abs(a).max())
