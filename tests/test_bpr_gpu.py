"""BPRMF parity: HIP backend (through the C ABI) vs the CPU oracle.

Tolerances (fp32; stated per BASELINE.json north_star "same factor matrices within a stated fp32
tolerance"): deterministic single-stream runs agree to max-abs error <= 1e-5 x max|value| after
several epochs (summation order inside a dot product is the only difference); gradient-accumulation
(adam / adagrad) runs agree to 1e-4 because atomics reorder sums; Hogwild runs are compared through
loss / NDCG@10."""
import numpy as np
import pytest

from conftest import bpr_opt, tiny_csr
import helpers as H

pytestmark = pytest.mark.gpu


def _factors(csr, d, vdim, seed=1, scale=0.3, bias=True):
    rng = np.random.default_rng(seed)
    P = H.pad(rng.normal(scale=scale, size=(csr.num_users, d)).astype(np.float32), vdim)
    Q = H.pad(rng.normal(scale=scale, size=(csr.num_items, d)).astype(np.float32), vdim)
    Qb = rng.normal(scale=0.1, size=(csr.num_items, 1)).astype(np.float32)
    if not bias:
        Qb *= 0
    return P, Q, Qb


def _vdim(d):
    return ((d + 31) // 32) * 32


DET = dict(sampler="counter", pos_order="csr", inline=True)


@pytest.mark.parametrize("d,kw", [
    (20, {}),
    (128, {}),
    (200, dict(num_negative_samples=2)),
    (64, dict(use_bias=False, verify_neg=False)),
    (40, dict(sampling_power=1.0)),
    (32, dict(update_j=False, reg_u=0.1)),
    (300, dict(update_i=False)),
])
def test_sequential_sgd_matches_oracle(oracle, d, kw):
    from buffalo_amd.backend import CyBPR
    csr = tiny_csr(U=40, I=60, density=0.2, seed=13)
    opt = bpr_opt(d=d, lr=0.05, min_lr=0.002, num_iters=3, random_seed=7, **kw)
    vdim = _vdim(d)
    P, Q, Qb = _factors(csr, d, vdim, bias=opt["use_bias"])
    Po, Qo, Qbo = P[:, :d].copy(), Q[:, :d].copy(), Qb.copy()
    H.run_oracle_sgd(oracle.OracleBPRMF, opt, csr, Po, Qo, Qbo, epochs=3, n_chunks=2, modes=DET)
    obj = H.run_hip_sgd(CyBPR, opt, csr, P, Q, Qb, epochs=3, n_chunks=2, modes=dict(sequential=1))
    assert H.relerr(P[:, :d], Po) < 1e-5, H.relerr(P[:, :d], Po)
    assert H.relerr(Q[:, :d], Qo) < 1e-5, H.relerr(Q[:, :d], Qo)
    assert H.relerr(Qb, Qbo) < 1e-5 or not opt["use_bias"]
    assert np.all(P[:, d:] == 0) and np.all(Q[:, d:] == 0)  # pad columns stay zero
    assert obj.stats()["samples"] == 3 * csr.nnz * opt["num_negative_samples"]


def test_reference_order_replay(oracle):
    """The oracle in pure reference mode (mt19937 + unordered_set order) records its (u,pos,neg)
    stream; replaying that stream through the HIP update path must land on the same model."""
    from buffalo_amd.backend import CyBPR
    csr = tiny_csr(U=50, I=70, density=0.15, seed=21)
    d, vdim = 24, 32
    opt = bpr_opt(d=d, lr=0.04, min_lr=0.04, num_iters=2, random_seed=777)
    P, Q, Qb = _factors(csr, d, vdim)
    Po, Qo, Qbo = P[:, :d].copy(), Q[:, :d].copy(), Qb.copy()
    o = H.run_oracle_sgd(oracle.OracleBPRMF, opt, csr, Po, Qo, Qbo, epochs=2, modes=dict(inline=True), trace=True)
    tr = o.get_trace()
    assert len(tr) == 2 * csr.nnz
    obj = CyBPR()
    path = H.write_opt(dict(opt, accelerator=True))
    assert obj.init(path)
    obj.set_mode("sequential", 1)
    obj.initialize_model(P, Q, Qb, csr.nnz, True)
    obj.update_triples(np.ascontiguousarray(tr[:, 0]), np.ascontiguousarray(tr[:, 1]), np.ascontiguousarray(tr[:, 2]), 0.04)
    obj.synchronize(True)
    assert H.relerr(P[:, :d], Po) < 1e-5 and H.relerr(Q[:, :d], Qo) < 1e-5 and H.relerr(Qb, Qbo) < 1e-5


@pytest.mark.parametrize("optimizer,pcn,kw,hip", [
    ("adagrad", False, {}, {}),
    ("adam", True, {}, {}),
    ("adam", False, {}, {}),
    ("adagrad", False, {}, dict(accum_two_pass=0)),                  # one atomic row add per triple and row
    ("adam", True, {}, dict(accum_two_pass=0)),
    ("adagrad", True, dict(update_j=False, use_bias=False), {}),    # only the positive list is gathered
    ("adam", False, dict(update_i=False, num_negative_samples=3, d=128), dict(n_chunks=3)),
    ("adagrad", False, dict(num_negative_samples=1, d=200), dict(resident=True)),   # cached positive list
])
def test_accumulate_modes_parallel(oracle, optimizer, pcn, kw, hip):
    """adam / adagrad freeze P,Q inside an epoch, so the fully parallel path must agree with the
    sequential oracle up to fp32 summation order (incl. Q-5, Q-6, Q-9).  Default: two passes -- the update kernel
    records (logit, negative) per triple and keeps gradP in registers, grad_gather_kernel sums the item-side rows
    over the item-sorted incidence lists; accum_two_pass=0: per-triple atomic row adds."""
    from buffalo_amd.backend import CyBPR
    csr = tiny_csr(U=64, I=80, density=0.2, seed=5)
    kw = dict(kw)
    d = kw.pop("d", 48)
    vdim = _vdim(d)
    kw.setdefault("num_negative_samples", 2)
    opt = bpr_opt(d=d, lr=0.03, num_iters=3, random_seed=3, optimizer=optimizer, per_coordinate_normalize=pcn, **kw)
    P, Q, Qb = _factors(csr, d, vdim, bias=opt["use_bias"])
    Po, Qo, Qbo = P[:, :d].copy(), Q[:, :d].copy(), Qb.copy()
    hip = dict(hip)
    n_chunks, resident = hip.pop("n_chunks", 1), hip.pop("resident", False)
    H.run_oracle_sgd(oracle.OracleBPRMF, opt, csr, Po, Qo, Qbo, epochs=3, n_chunks=n_chunks, modes=DET)
    H.run_hip_sgd(CyBPR, opt, csr, P, Q, Qb, epochs=3, n_chunks=n_chunks, resident=resident, modes=dict(chunk=64, **hip))
    assert H.relerr(P[:, :d], Po) < 1e-4, H.relerr(P[:, :d], Po)
    assert H.relerr(Q[:, :d], Qo) < 1e-4, H.relerr(Q[:, :d], Qo)
    assert H.relerr(Qb, Qbo) < 1e-4


def test_triples_parallel_disjoint_rows(oracle):
    """Conflict-free triples (every row touched once) give the same result whatever the schedule."""
    from buffalo_amd.backend import CyBPR
    U, I, d, vdim = 300, 600, 128, 128
    rng = np.random.default_rng(0)
    users = rng.permutation(U).astype(np.int32)
    items = rng.permutation(I).astype(np.int32)
    pos, neg = np.ascontiguousarray(items[:U]), np.ascontiguousarray(items[U:2 * U])
    opt = bpr_opt(d=d, lr=0.1, min_lr=0.1)
    outs = []
    for mode in (dict(sequential=1), dict(hogwild_atomic=1, chunk=64), dict(hogwild_atomic=0, chunk=64, prefetch=0),
                 dict(hogwild_atomic=2, xcd_hot_tau=0), dict(hogwild_atomic=2, xcd_hot_tau=0, xcd_sync_updates=64),
                 dict(hogwild_atomic=2, xcd_hot_tau=1), dict(hogwild_atomic=2, xcd_hot_tau=100, xcd_merge_mean=0, chunk=128),
                 dict(hogwild_atomic=2, xcd_hot_tau=0, xcd_fresh=1), dict(hogwild_atomic=2, xcd_hot_tau=0, xcd_v4=1),
                 dict(hogwild_atomic=2, xcd_hot_tau=1, xcd_v4=1, xcd_fresh=1)):
        rng2 = np.random.default_rng(1)
        P = rng2.normal(scale=0.3, size=(U, vdim)).astype(np.float32)
        Q = rng2.normal(scale=0.3, size=(I, vdim)).astype(np.float32)
        Qb = rng2.normal(scale=0.1, size=(I, 1)).astype(np.float32)
        obj = CyBPR()
        assert obj.init(H.write_opt(dict(opt, accelerator=True)))
        for k, v in mode.items():
            obj.set_mode(k, v)
        obj.initialize_model(P, Q, Qb, U, True)
        obj.update_triples(users, pos, neg, 0.1)
        obj.synchronize(True)
        outs.append((P, Q, Qb))
    for a in outs[1:]:
        for x, y in zip(a, outs[0]):
            assert H.relerr(x, y) < 2e-6


@pytest.mark.parametrize("d,kw,modes", [
    (128, {}, {}),
    (40, {}, dict(im_presample=0, im_blocks=1)),            # negatives drawn inside the walk, one run per item and queue
    (128, {}, dict(xcd_hot_tau=1)),                      # every user / negative row "hot": the atomic paths
    (64, {}, dict(im_drain_only=1)),                     # the any-XCD drain launch does all the work
    (300, dict(num_negative_samples=2), dict(prefetch=1)),  # two-triples-ahead prefetch slots + re-read before store
    (200, dict(num_negative_samples=3), dict(prefetch=1, xcd_fresh=0, im_presample=0)),
    (128, dict(num_negative_samples=2, use_bias=False), dict(im_max_stale=1)),
    (96, dict(update_j=False), dict(xcd_sync_updates=1024)),
    (32, dict(update_i=False), {}),
    (64, {}, dict(n_chunks=3)),                          # the reference's call pattern: keys sent chunk by chunk, nothing resident
    (128, dict(num_negative_samples=2), dict(im_user_replicas=0)),       # one owner XCD per user (the big-shard form) on this small matrix
    (128, dict(num_negative_samples=2), dict(im_user_replicas=1, xcd_sync_updates=1024)),   # per-XCD replicas of P, several merges
    (64, {}, dict(im_user_replicas=1, n_chunks=3)),      # replicas over a row range per call
    (128, {}, dict(im_dual=0)),                          # one triple per wave at vdim <= 128 (the default there is two)
    (128, dict(num_negative_samples=2), dict(im_dual=0, xcd_hot_tau=1)),
    (128, {}, dict(im_dual=1)),                          # two triples per wave (bpr_item_major_dual_kernel)
    (96, dict(num_negative_samples=3), dict(im_dual=1, xcd_sync_updates=1024)),
    (128, dict(num_negative_samples=2, use_bias=False), dict(im_dual=1, im_max_stale=1, xcd_hot_tau=1)),   # flush every triple, atomic rows
    (40, dict(update_j=False), dict(im_dual=1, im_presample=0, im_user_replicas=1)),
    (32, dict(update_i=False), dict(im_dual=1, n_chunks=3)),
    (64, dict(num_negative_samples=2), dict(im_dual=1)),                  # round 6: the whole-group instantiation NK = 2 (128 / 96 / 32 above: NK = 4 / 3 / 1)
    (128, {}, dict(im_dual=1, im_dual_generic=1)),       # ... and the guarded instantiation at a vdim the whole-group one normally takes
    # lr <= 0.01: the heavy users alone get per-XCD replicas of P (im_user_hybrid, the whole-matrix default).  Every user of this
    # matrix has one entry, so with the collision threshold at 0.5 all of them count as heavy at the owner's share and none at the
    # spread share: every row goes through a replica and the delta-rule merge; with the knob off through its owner XCD
    (128, dict(lr=0.005, min_lr=0.005), dict(xcd_hot_tau=500)),
    (128, dict(lr=0.005, min_lr=0.005, num_negative_samples=2), dict(xcd_hot_tau=500, im_dual=0, xcd_sync_updates=1024, im_user_hybrid=1)),
    (96, dict(lr=0.005, min_lr=0.005), dict(xcd_hot_tau=500, im_user_hybrid=0)),
    # the merges' saturation weights (round 4; bias weight on by default): a row that saw at most one step is merged with weight 1 exactly
    (128, dict(num_negative_samples=2), dict(xcd_stiff_q=1000, xcd_stiff_b=1000)),
    (128, dict(lr=0.005, min_lr=0.005), dict(xcd_hot_tau=500, xcd_stiff_p=1000, xcd_stiff_q=250)),
    (128, {}, dict(xcd_stiff_b=0)),
])
def test_item_major_conflict_free(oracle, d, kw, modes):
    """hogwild_atomic=3 (item-major walk, users owned by XCDs, Q[i] in registers, Q[j] in per-XCD replicas):
    on a matrix where every user has one positive and the positives are distinct items, a triple whose
    three rows no other triple touches has one possible result -- the sequential oracle's (same counter
    sampler), whatever the schedule, the queue a wave serves or the path (plain / atomic / drain) a row takes."""
    from buffalo_amd import synth
    from buffalo_amd.backend import CyBPR
    U, I = 3000, 60000
    rng = np.random.default_rng(5)
    keys = rng.permutation(I)[:U].astype(np.int32)
    csr = synth.CSR(U, I, np.arange(1, U + 1, dtype=np.int64), keys, np.ones(U, np.float32))
    opt = bpr_opt(**dict(dict(d=d, lr=0.05, min_lr=0.05, num_iters=1, random_seed=11), **kw))
    vdim = _vdim(d)
    P, Q, Qb = _factors(csr, d, vdim, bias=opt["use_bias"])
    P0, Q0 = P.copy(), Q.copy()
    Po, Qo, Qbo = P[:, :d].copy(), Q[:, :d].copy(), Qb.copy()
    o = H.run_oracle_sgd(oracle.OracleBPRMF, opt, csr, Po, Qo, Qbo, epochs=1, modes=DET, trace=True)
    tr = o.get_trace()
    nn = opt["num_negative_samples"]
    assert len(tr) == U * nn
    modes = dict(modes)
    n_chunks = modes.pop("n_chunks", 1)
    obj = H.run_hip_sgd(CyBPR, opt, csr, P, Q, Qb, epochs=1, n_chunks=n_chunks, modes=dict(hogwild_atomic=3, **modes), resident=n_chunks == 1)
    st = obj.stats()
    assert st["samples"] == U * nn and st["merges"] >= 1
    # rows touched by exactly one entry
    touch = np.zeros(I, np.int64)
    np.add.at(touch, keys, 1)                       # each entry's positive once
    np.add.at(touch, tr[:, 2], 1)                   # every drawn negative
    ent_neg_clean = (touch[tr[:, 2]] == 1).reshape(U, nn).all(axis=1)
    clean = (touch[keys] == 1) & ent_neg_clean
    assert clean.sum() > U // 4
    cu = np.flatnonzero(clean)
    ci = keys[cu]
    cj = tr[:, 2].reshape(U, nn)[cu].reshape(-1)
    assert H.relerr(P[cu][:, :d], Po[cu]) < 1e-5, H.relerr(P[cu][:, :d], Po[cu])
    assert H.relerr(Q[ci][:, :d], Qo[ci]) < 1e-5, H.relerr(Q[ci][:, :d], Qo[ci])
    assert H.relerr(Q[cj][:, :d], Qo[cj]) < 1e-5, H.relerr(Q[cj][:, :d], Qo[cj])
    if opt["use_bias"]:
        assert H.relerr(Qb[ci], Qbo[ci]) < 1e-5 and H.relerr(Qb[cj], Qbo[cj]) < 1e-5
    # something was learned, nothing else moved, pad columns stay zero
    assert not np.array_equal(P[cu], P0[cu])
    untouched = np.flatnonzero(touch == 0)
    np.testing.assert_array_equal(Q[untouched], Q0[untouched])
    assert np.isfinite(P).all() and np.isfinite(Q).all() and np.all(P[:, d:] == 0) and np.all(Q[:, d:] == 0)


_item_major_order = H.item_major_order


@pytest.mark.parametrize("d,nn,blocks", [(48, 2, 3), (128, 1, 1), (200, 3, 8)])
def test_item_major_single_wave_equals_sequential_replay(oracle, d, nn, blocks):
    """Whole epochs of the item-major kernel against the ORACLE's arithmetic: one wave draining all queues applies the
    epoch's triples in a known order (bfh_bpr_item_major_plan); the oracle's own triples of that epoch (same counter
    sampler), re-ordered that way and pushed through `oracle.apply_triples` -- the loop body of bpr.cc:119-171 over an
    explicit list -- must give the same model.

    Why this is a tolerance and not bit equality, quantified: the kernel sums a dot product per lane (float4 columns)
    and then across lanes, the oracle left to right (omp simd), so x_uij differs in its last bits -- and the sigmoid
    table index `(int)((x + 6) * 83)` (Q-2) is discontinuous in x.  The kernel records the table index of every triple
    (test hook "im_trace") and the replay records its own, so the comparison is made at the level of the individual
    step: the first triple whose float32 score rounds across a table boundary takes a neighbouring entry (a logit that
    differs by <= 0.003), the models separate by ~1e-4, and from then on every score within that distance of a boundary
    flips too.  Measured (profiles/r02_gpu_tests.txt): 0.6-2.3 % of the triples take another entry, never more than 5
    entries away, and the models end 0.5-1.2e-3 apart.  Asserted: >= 97 % identical table indices, none further than
    8 entries (a misplaced triple, a stale row or a wrong negative lands anywhere in the 1000-entry table and moves the
    model by 5-12 %: the CSR-order model below), model distance < 3e-3."""
    import torch
    from buffalo_amd.backend import CyBPR
    import ref_numpy as R
    csr = tiny_csr(U=300, I=200, density=0.08, seed=3)
    vdim = _vdim(d)
    lr = 0.05
    opt = bpr_opt(d=d, lr=lr, min_lr=lr, num_iters=2, random_seed=5, num_negative_samples=nn)
    P, Q, Qb = _factors(csr, d, vdim)
    P0 = P.copy()
    Po, Qo, Qbo = P[:, :d].copy(), Q[:, :d].copy(), Qb.copy()           # the oracle's own CSR-order epochs (for the triples)
    Pr, Qr, Qbr = P[:, :d].copy(), Q[:, :d].copy(), Qb.copy()           # the replay through oracle.apply_triples
    o = H.run_oracle_sgd(oracle.OracleBPRMF, opt, csr, Po, Qo, Qbo, epochs=2, modes=DET, trace=True)
    tr = o.get_trace()
    n = csr.nnz * nn
    assert len(tr) == 2 * n
    order = _item_major_order(csr, 8, blocks, nn)
    assert np.array_equal(np.sort(order), np.arange(n))
    rep = oracle.OracleBPRMF()
    assert rep.init(H.write_opt(opt))
    rep.initialize_model(Pr, Qr, Qbr, csr.nnz)
    # the item-major kernel, one wave, the same two epochs, recording its table indices
    obj = CyBPR()
    assert obj.init(H.write_opt(dict(opt, accelerator=True)))
    for k, v in dict(hogwild_atomic=3, im_single_wave=1, im_force_queues=8, im_blocks=blocks, xcd_sync_updates=1 << 40, im_trace=n).items():
        obj.set_mode(k, v)
    obj.initialize_model(P, Q, Qb, csr.nnz, True)
    obj.set_cumulative_table(H.cum_table(csr, opt), csr.num_items)
    obj.set_resident_csr(csr.indptr, csr.keys)
    table = R.exp_table()
    flips = worst = 0
    for e in range(2):
        te = np.ascontiguousarray(tr[e * n:(e + 1) * n][order])
        obj.add_jobs(0, csr.num_users, csr.indptr, None)
        obj.update_parameters()
        idx_hip = obj.device_tensor("im_trace", (n,), dtype="int32").cpu().numpy()
        torch.cuda.synchronize()
        # the oracle's model BEFORE each of its steps gives the index the reference arithmetic takes (numpy float32, same loop)
        Pn, Qn, Qbn = Pr.copy(), Qr.copy(), Qbr.copy()
        idx_ref = np.empty(n, np.int32)
        for t, (u, i, j) in enumerate(te):
            x = np.float32(np.float32(np.dot(Pn[u], Qn[i] - Qn[j])) + np.float32(Qbn[i, 0] - Qbn[j, 0]))
            idx_ref[t] = 1000 if x > 6 else (-1 if x < -6 else int(np.float32(x + np.float32(6)) * np.float32(83)))
            R.bpr_sgd_step(Pn, Qn, Qbn, u, i, j, lr, opt, table)
        flips += int((idx_hip != idx_ref).sum())
        worst = max(worst, int(np.abs(idx_hip - idx_ref).max()))
        rep.apply_triples(np.ascontiguousarray(te[:, 0]), np.ascontiguousarray(te[:, 1]), np.ascontiguousarray(te[:, 2]), lr)
    obj.synchronize(True)
    assert not np.array_equal(Pr, P0[:, :d])                      # the replay moved the model
    step = float(np.abs(np.diff(table)).max())                    # largest logit increment between two table entries
    tol = 3e-3
    errs = (H.relerr(P[:, :d], Pr), H.relerr(Q[:, :d], Qr), H.relerr(Qb, Qbr))
    print("\nitem-major replay d=%d nn=%d blocks=%d: %d triples, %d took another table entry (at most %d entries away), max table "
          "increment %.2e, tol %.2e, errors %s" % (d, nn, blocks, 2 * n, flips, worst, step, tol, errs))
    assert worst <= 8                                               # a different triple would land anywhere in the 1000-entry table
    assert flips <= 0.03 * 2 * n                                    # >= 97 % of the steps take the very same table entry
    for err in errs:
        assert err < tol, (err, tol, flips)
    # the order matters far more than that: the CSR-order model is somewhere else (the comparison is not vacuous)
    assert H.relerr(P[:, :d], Po) > 10 * max(tol, 1e-3) and H.relerr(Pr, Po) > 10 * max(tol, 1e-3)


def test_compute_loss_matches_oracle(oracle):
    from buffalo_amd.backend import CyBPR
    csr = tiny_csr(U=30, I=40, seed=2)
    d, vdim = 20, 32
    opt = bpr_opt(d=d)
    P, Q, Qb = _factors(csr, d, vdim, scale=1.0)
    rng = np.random.default_rng(4)
    u = rng.integers(0, 30, 17).astype(np.int32)
    i = rng.integers(0, 40, 17).astype(np.int32)
    j = rng.integers(0, 40, 17).astype(np.int32)
    o = oracle.OracleBPRMF()
    assert o.init(H.write_opt(opt))
    Po, Qo = P[:, :d].copy(), Q[:, :d].copy()
    o.initialize_model(Po, Qo, Qb.copy(), csr.nnz)
    want = o.compute_loss(u, i, j)
    obj = CyBPR()
    assert obj.init(H.write_opt(dict(opt, accelerator=True)))
    obj.initialize_model(P, Q, Qb, csr.nnz, True)
    got = obj.compute_loss(u, i, j)
    assert abs(got - want) < 1e-5 * max(1.0, abs(want))


@pytest.mark.parametrize("atomic", [1, 0, 2, 3, 30, 31, 32])
def test_hogwild_statistical_parity(oracle, atomic):
    """Throughput mode vs the threaded reference path: same ranking quality on planted low-rank data
    (mirrors the ndcg threshold test, tests/algo/test_bpr.py:38-47).  With fp32 atomics no update is
    lost and the result must track the reference; write-through racy stores (hogwild_atomic=0) drop
    colliding updates -- hundreds of waves hit a 400-item table in lock-step -- so that mode only has
    to learn something (it is an opt-in)."""
    from buffalo_amd import synth
    from buffalo_amd.backend import CyBPR
    csr, vali = synth.planted(600, 400, d_true=6, density=0.06, seed=7)
    d, vdim = 16, 32
    opt = bpr_opt(d=d, lr=0.05, min_lr=0.01, num_iters=30, random_seed=7, num_workers=4, reg_u=0.01, reg_i=0.01, reg_j=0.01,
                  reg_b=0.01)
    P0, Q0, Qb0 = synth.init_factors(600, 400, d, seed=7)
    Po, Qo, Qbo = P0.copy(), Q0.copy(), Qb0.copy()
    H.run_oracle_sgd(oracle.OracleBPRMF, opt, csr, Po, Qo, Qbo, epochs=30)
    P, Q, Qb = H.pad(P0, vdim), H.pad(Q0, vdim), Qb0.copy()
    extra = {}
    if atomic == 32:      # item-major, one triple per wave (vdim 32: the default is two)
        extra["im_dual"] = 0
        atomic = 3
    elif atomic >= 30:    # item-major with users owned by one XCD each (30) / with per-XCD replicas of P (31); 3 = by shard size
        extra["im_user_replicas"] = atomic - 30
        atomic = 3
    H.run_hip_sgd(CyBPR, opt, csr, P, Q, Qb, epochs=30, modes=dict(hogwild_atomic=atomic, chunk=64, **extra), resident=True)
    n_ref = H.ndcg_at_k(Po, Qo, csr, vali, Qb=Qbo)
    n_hip = H.ndcg_at_k(P[:, :d], Q[:, :d], csr, vali, Qb=Qb)
    base = H.ndcg_at_k(P0, Q0, csr, vali, Qb=Qb0)
    assert np.isfinite(P).all() and np.isfinite(Q).all()
    assert n_ref > 3 * max(base, 0.01)
    print("hogwild_atomic=%d ndcg %.4f (reference path %.4f, untrained %.4f)" % (atomic, n_hip, n_ref, base))
    # 1: atomics everywhere; 2: per-XCD replicas + atomics on the rows the popularity rule marks hot; 3: item-major
    if atomic:
        assert n_hip > 3 * max(base, 0.01)
        assert abs(n_hip - n_ref) < 0.25 * n_ref, (n_hip, n_ref)
    else:
        assert n_hip > base, (n_hip, base)


_FULL = {}


def _full_size_csr():
    from buffalo_amd import synth
    if "csr" not in _FULL:
        import bench   # same generator call, cached under /tmp for the other full-size runs on this box
        _FULL["csr"] = bench.load_matrix("ml20m", 7)
    return _FULL["csr"]


@pytest.mark.parametrize("policy", [1, 2, 3])
def test_full_size_properties(policy):
    """BASELINE config #2 shape (138,493 x 27,278, 20,000,263 nnz, d=128): size-independent checks, for the
    atomic policy and for the per-XCD replica policy (several merges per epoch, hot rows on atomics)."""
    from buffalo_amd import synth
    from buffalo_amd.backend import CyBPR
    U, I, nnz = synth.SHAPES["ml20m"]
    csr = _full_size_csr()
    d = vdim = 128
    opt = bpr_opt(d=d, lr=0.0, min_lr=0.0, num_iters=2, random_seed=7, compute_loss_on_training=True)
    P, Q, Qb = synth.init_factors(U, I, d, seed=7)
    P0, Q0, Qb0 = P.copy(), Q.copy(), Qb.copy()
    obj = CyBPR()
    assert obj.init(H.write_opt(dict(opt, accelerator=True)))
    obj.set_mode("hogwild_atomic", policy)
    obj.initialize_model(P, Q, Qb, nnz, True)
    obj.set_cumulative_table(np.zeros(I, np.int64), I)
    obj.set_resident_csr(csr.indptr, csr.keys)
    # (a) lr == 0: one epoch is the identity, bit for bit, and every nnz is visited exactly once
    loss0, n = obj.add_jobs(0, U, csr.indptr, None)
    obj.update_parameters()
    assert n == nnz and obj.stats()["samples"] == nnz
    assert (obj.stats()["merges"] >= 2) == (policy >= 2)
    np.testing.assert_array_equal(P, P0)
    np.testing.assert_array_equal(Q, Q0)
    np.testing.assert_array_equal(Qb, Qb0)
    # the sampled training loss at init is ~ log(2) per sample (scores ~ 0)
    assert abs(loss0 / n - np.log(2.0)) < 1e-2
    # (b) a real epoch lowers the sampled loss and keeps everything finite
    obj2 = CyBPR()
    opt2 = dict(opt, lr=0.05, min_lr=0.05, accelerator=True)
    assert obj2.init(H.write_opt(opt2))
    obj2.set_mode("hogwild_atomic", policy)
    obj2.initialize_model(P, Q, Qb, nnz, True)
    obj2.set_cumulative_table(np.zeros(I, np.int64), I)
    obj2.set_resident_csr(csr.indptr, csr.keys)
    l1, _ = obj2.add_jobs(0, U, csr.indptr, None)
    obj2.update_parameters()
    l2, _ = obj2.add_jobs(0, U, csr.indptr, None)
    obj2.update_parameters()
    # (the item-major walk adapts an item's row within its run of triples, so its sampled loss starts lower)
    assert l2 < l1 * (0.98 if policy != 3 else 1.0), (l1, l2)
    assert np.isfinite(P).all() and np.isfinite(Q).all() and np.isfinite(Qb).all()
    assert not np.array_equal(P, P0) and not np.array_equal(Q, Q0)
