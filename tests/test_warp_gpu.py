"""WARP parity: HIP backend vs the CPU oracle.  P and Q are frozen inside an epoch, so with the
shared counter sampler the parallel kernel must reproduce the oracle's accept/reject decisions and
gradients exactly up to fp32 summation order (tolerance 1e-4 x max|value| after 3 epochs)."""
import numpy as np
import pytest

from conftest import tiny_csr, warp_opt
import helpers as H

pytestmark = pytest.mark.gpu

DET = dict(sampler="counter", pos_order="csr", inline=True)


def _run_pair(oracle, csr, d, opt, epochs, scale, modes=None, seed=3):
    from buffalo_amd.backend import CyWARP
    vdim = ((d + 31) // 32) * 32
    rng = np.random.default_rng(seed)
    P = H.pad(rng.normal(scale=scale, size=(csr.num_users, d)).astype(np.float32), vdim)
    Q = H.pad(rng.normal(scale=scale, size=(csr.num_items, d)).astype(np.float32), vdim)
    Qb = np.zeros((csr.num_items, 1), np.float32)
    Po, Qo = P[:, :d].copy(), Q[:, :d].copy()
    o = H.run_oracle_sgd(oracle.OracleWARP, opt, csr, Po, Qo, Qb.copy(), epochs=epochs, n_chunks=2, modes=DET)
    obj = H.run_hip_sgd(CyWARP, opt, csr, P, Q, Qb, epochs=epochs, n_chunks=2, modes=modes or {})
    return o, obj, (P, Q), (Po, Qo)


@pytest.mark.parametrize("d,kw,scale", [
    (20, dict(max_trials=10, threshold=0.3), 0.5),
    (64, dict(max_trials=500, threshold=1.0), 0.3),
    (256, dict(max_trials=30, threshold=0.5, reg_u=0.01, reg_i=0.02, reg_j=0.03), 0.1),
    (40, dict(max_trials=8, threshold=0.2, optimizer="adam", per_coordinate_normalize=True), 0.4),
    (32, dict(max_trials=20, threshold=0.5, score_func="l2"), 0.4),
    (96, dict(max_trials=12, threshold=0.4, score_func="l2", reg_i=0.02, reg_j=0.01, per_coordinate_normalize=True), 0.3),
])
@pytest.mark.parametrize("two_pass", [1, 0])
def test_epochs_match_oracle(oracle, d, kw, scale, two_pass):
    """two_pass=1 (default): the item-side gradient rows are summed by the sorted gather (grad_gather_kernel);
    two_pass=0: one atomic row add per accepted positive and row."""
    csr = tiny_csr(U=48, I=90, density=0.12, seed=17)
    opt = warp_opt(d=d, random_seed=11, num_iters=3, lr=0.05, **kw)
    o, obj, (P, Q), (Po, Qo) = _run_pair(oracle, csr, d, opt, 3, scale, modes=dict(chunk=64, accum_two_pass=two_pass))
    so, sg = o.stats(), obj.stats()
    assert sg["scored_negatives"] == so["scored_negatives"]      # identical trial sequences (Q-10)
    assert sg["accepted"] == so["updates"]
    assert 0 < sg["accepted"] <= 3 * csr.nnz
    assert H.relerr(P[:, :d], Po) < 1e-4, H.relerr(P[:, :d], Po)
    assert H.relerr(Q[:, :d], Qo) < 1e-4, H.relerr(Q[:, :d], Qo)
    assert np.all(P[:, d:] == 0) and np.all(Q[:, d:] == 0)
    # Q-12: every row ends inside the unit ball
    assert np.linalg.norm(P, axis=1).max() <= 1 + 1e-5 and np.linalg.norm(Q, axis=1).max() <= 1 + 1e-5


def test_sequential_equals_parallel(oracle):
    csr = tiny_csr(U=40, I=64, density=0.15, seed=23)
    opt = warp_opt(d=64, random_seed=5, num_iters=2, max_trials=16, threshold=0.4)
    outs = []
    for modes in (dict(sequential=1, accum_two_pass=0, warp_presample=0), dict(sequential=1), dict(chunk=64), dict(chunk=256, waves_per_cu=4),
                  dict(chunk=64, accum_two_pass=0), dict(chunk=64, warp_presample=0), dict(chunk=128, warp_presample=8),
                  dict(chunk=64, warp_presample=4, accum_two_pass=0)):
        _, obj, (P, Q), _ = _run_pair(oracle, csr, 64, opt, 2, 0.3, modes=modes)
        outs.append((P, Q, obj.stats()["scored_negatives"]))
    for P, Q, sc in outs[1:]:
        assert sc == outs[0][2]
        assert H.relerr(P, outs[0][0]) < 1e-5 and H.relerr(Q, outs[0][1]) < 1e-5


def test_compute_loss_matches_oracle(oracle):
    from buffalo_amd.backend import CyWARP
    d, vdim = 30, 32
    rng = np.random.default_rng(9)
    P = H.pad(rng.normal(size=(20, d)).astype(np.float32), vdim)
    Q = H.pad(rng.normal(size=(35, d)).astype(np.float32), vdim)
    Qb = np.zeros((35, 1), np.float32)
    u = rng.integers(0, 20, 64).astype(np.int32)
    i = rng.integers(0, 35, 64).astype(np.int32)
    j = rng.integers(0, 35, 64).astype(np.int32)
    opt = warp_opt(d=d, threshold=1.5)
    o = oracle.OracleWARP()
    assert o.init(H.write_opt(opt))
    o.initialize_model(P[:, :d].copy(), Q[:, :d].copy(), Qb.copy(), 100)
    obj = CyWARP()
    assert obj.init(H.write_opt(dict(opt, accelerator=True)))
    obj.initialize_model(P, Q, Qb, 100, True)
    assert obj.compute_loss(u, i, j) == o.compute_loss(u, i, j)


def test_training_improves_ranking(oracle):
    """Mirror of tests/algo/test_warp.py:44-48 on planted data, both backends."""
    from buffalo_amd import synth
    from buffalo_amd.backend import CyWARP
    csr, vali = synth.planted(500, 300, d_true=6, density=0.06, seed=3)
    d, vdim = 24, 32
    opt = warp_opt(d=d, random_seed=7, num_iters=15, lr=0.05, num_workers=4)
    P0, Q0, Qb0 = synth.init_factors(500, 300, d, seed=7, signed=True)
    Qb0 *= 0
    Po, Qo = P0.copy(), Q0.copy()
    H.run_oracle_sgd(oracle.OracleWARP, opt, csr, Po, Qo, Qb0.copy(), epochs=15)
    P, Q = H.pad(P0, vdim), H.pad(Q0, vdim)
    H.run_hip_sgd(CyWARP, opt, csr, P, Q, Qb0.copy(), epochs=15, resident=True)
    n_ref = H.ndcg_at_k(Po, Qo, csr, vali)
    n_hip = H.ndcg_at_k(P[:, :d], Q[:, :d], csr, vali)
    base = H.ndcg_at_k(P0, Q0, csr, vali)
    assert n_ref > 3 * max(base, 0.01) and n_hip > 3 * max(base, 0.01), (base, n_ref, n_hip)
    assert abs(n_hip - n_ref) < 0.25 * n_ref, (n_hip, n_ref)
