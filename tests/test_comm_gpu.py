"""The library's own communicator (bfh_comm_*, RCCL) on ONE GPU: a world of one rank exercises everything but the wire --
RCCL is loaded and initialised, the exchange kernels, streams, events and the pipelining run, and the sums over "all ranks"
are the rank's own contribution, so the results must equal the runs without a communicator.  (The protocol across ranks is
tested under gloo on CPU: tests/test_dist_cpu.py; RCCL refuses two ranks on one device, so >1 rank needs >1 GPU.)"""
import numpy as np
import pytest

from conftest import als_opt, bpr_opt, tiny_csr, warp_opt
import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm():
    from buffalo_amd.backend import Comm
    c = Comm(1, 0, Comm.unique_id(), 0)
    c.self_test()
    assert c.all_reduce([1.5, -2.0]) == [1.5, -2.0]
    return c


def _factors(csr, d, vdim, seed=1):
    rng = np.random.default_rng(seed)
    P = H.pad(rng.normal(scale=0.3, size=(csr.num_users, d)).astype(np.float32), vdim)
    Q = H.pad(rng.normal(scale=0.3, size=(csr.num_items, d)).astype(np.float32), vdim)
    Qb = rng.normal(scale=0.1, size=(csr.num_items, 1)).astype(np.float32)
    return P, Q, Qb


def _run(cls, opt, csr, P, Q, Qb, epochs, comm, modes, n_chunks=1):
    obj = cls()
    assert obj.init(H.write_opt(dict(opt, accelerator=True)))
    for k, v in modes.items():
        obj.set_mode(k, v)
    obj.initialize_model(P, Q, Qb, csr.nnz, True)
    obj.set_cumulative_table(H.cum_table(csr, opt), csr.num_items)
    obj.set_resident_csr(csr.indptr, csr.keys)
    if comm is not None:
        obj.set_comm(comm)
    for _ in range(epochs):
        for a, b in H.chunks_of(csr, n_chunks):
            obj.add_jobs(a, b, csr.indptr, None)
        obj.update_parameters()
    st = obj.stats()
    obj.synchronize(True)
    obj.set_comm(None)
    return st


@pytest.mark.parametrize("modes,n_chunks", [
    (dict(sequential=1), 1),                                   # user-major walk: exchange after every call
    (dict(sequential=1, comm_overlap=0), 2),
    # item-major: one merge segment = one exchange point per call, left in flight / finished before the call returns
    (dict(hogwild_atomic=3, im_single_wave=1, im_force_queues=4, comm_segments=1), 1),
    (dict(hogwild_atomic=3, im_single_wave=1, im_force_queues=4, comm_segments=1, comm_overlap=0), 2),
    # three exchange segments per call; the run without a communicator is given the merge interval that cuts the call the same way
    (dict(hogwild_atomic=3, im_single_wave=1, im_force_queues=4, comm_segments=3, xcd_sync_updates="nnz/3"), 1),
])
def test_sgd_one_rank_world_equals_no_comm(comm, modes, n_chunks):
    """Deterministic walks: with one rank R == S, so folding "the other ranks' part" in adds exactly zero and a flush
    writes Q <- Z + S = the rank's own Q: bit-identical to the run without a communicator."""
    from buffalo_amd.backend import CyBPR
    csr = tiny_csr(U=60, I=80, density=0.15, seed=4)
    d, vdim = 40, 64
    opt = bpr_opt(d=d, lr=0.05, min_lr=0.01, num_iters=3, random_seed=5)
    modes = dict(modes)
    if modes.get("xcd_sync_updates") == "nnz/3":
        modes["xcd_sync_updates"] = (csr.nnz + 2) // 3
    ref = _factors(csr, d, vdim)
    got = tuple(a.copy() for a in ref)
    _run(CyBPR, opt, csr, *ref, 3, None, modes, n_chunks)
    st = _run(CyBPR, opt, csr, *got, 3, comm, modes, n_chunks)
    assert st["exchanges"] >= 3 * n_chunks
    for a, b in zip(got, ref):
        assert H.relerr(a, b) < 2e-6, H.relerr(a, b)     # Z + (Q - Z) rounds once per exchange


@pytest.mark.parametrize("modes,lo,hi", [
    ({}, 30, 30),                                 # default: a call is one segment, its exchange blocking
    (dict(comm_segments=4), 30 * 4, 30 * 5),      # pinned: 4 pipelined exchange segments per call (5 when the plan rounds up)
])
def test_item_major_default_with_comm_runs_and_learns(comm, modes, lo, hi):
    from buffalo_amd import synth
    from buffalo_amd.backend import CyBPR
    csr, vali = synth.planted(600, 400, d_true=6, density=0.06, seed=7)
    d, vdim = 16, 32
    opt = bpr_opt(d=d, lr=0.05, min_lr=0.01, num_iters=30, random_seed=7, reg_u=0.01, reg_i=0.01, reg_j=0.01, reg_b=0.01)
    P0, Q0, Qb0 = synth.init_factors(600, 400, d, seed=7)
    P, Q, Qb = H.pad(P0, vdim), H.pad(Q0, vdim), Qb0.copy()
    st = _run(CyBPR, opt, csr, P, Q, Qb, 30, comm, modes)
    assert lo <= st["exchanges"] <= hi, st["exchanges"]
    base = H.ndcg_at_k(P0, Q0, csr, vali, Qb=Qb0)
    assert np.isfinite(P).all() and np.isfinite(Q).all()
    assert H.ndcg_at_k(P[:, :d], Q[:, :d], csr, vali, Qb=Qb) > 3 * max(base, 0.01)


@pytest.mark.parametrize("cls_name,opt", [
    ("CyBPR", bpr_opt(d=48, lr=0.03, num_iters=3, random_seed=3, optimizer="adagrad", num_negative_samples=2)),
    ("CyBPR", bpr_opt(d=48, lr=0.03, num_iters=3, random_seed=3, optimizer="adam", per_coordinate_normalize=True)),
    ("CyWARP", warp_opt(d=64, random_seed=11, num_iters=3, lr=0.05, max_trials=30, threshold=0.5, reg_i=0.02)),
])
def test_gradient_exchange_one_rank_world(comm, cls_name, opt):
    """adam / adagrad / WARP: update_parameters sums the gradient deltas over the ranks first; with one rank gradQ = Z + (gradQ - Z)."""
    import buffalo_amd.backend as B
    csr = tiny_csr(U=64, I=80, density=0.2, seed=5)
    d = opt["d"]
    vdim = ((d + 31) // 32) * 32
    ref = _factors(csr, d, vdim)
    if cls_name == "CyWARP":
        ref[2][:] = 0
    got = tuple(a.copy() for a in ref)
    _run(getattr(B, cls_name), opt, csr, *ref, 3, None, dict(chunk=64))
    st = _run(getattr(B, cls_name), opt, csr, *got, 3, comm, dict(chunk=64))
    assert st["exchanges"] == 3
    for a, b in zip(got, ref):
        assert H.relerr(a, b) < 1e-4, H.relerr(a, b)


def test_als_publish_rows_one_rank_world(comm, oracle):
    from buffalo_amd.backend import CyALS
    from buffalo_amd.dist import CommDataParallelALS
    csr = tiny_csr(U=70, I=45, density=0.2, seed=11, counts=True)
    t = csr.transpose()
    rng = np.random.default_rng(2)
    opt = als_opt(d=32, num_iters=2, alpha=4.0, accelerator=True)
    outs = []
    for use_comm in (False, True):
        P = np.abs(rng.normal(scale=0.2, size=(70, 32))).astype(np.float32) if not outs else P0.copy()
        Q = np.abs(rng.normal(scale=0.2, size=(45, 32))).astype(np.float32) if not outs else Q0.copy()
        if not outs:
            P0, Q0 = P.copy(), Q.copy()
        g = CyALS()
        assert g.init(H.write_opt(opt))
        g.initialize_model(P, Q)
        g.set_resident_csr(0, csr.indptr, csr.keys, csr.vals)
        g.set_resident_csr(1, t.indptr, t.keys, t.vals)
        if use_comm:
            g.set_comm(comm)
            dp = CommDataParallelALS(g, comm, (csr.indptr, t.indptr), 70, 45)
            for _ in range(2):
                dp.epoch()
            assert g.stats()["exchanges"] == 4
        else:
            g.set_mode("als_writeback", 0)
            for _ in range(2):
                for axis, m in ((0, csr), (1, t)):
                    g.precompute(axis)
                    g.partial_update(0, m.num_users, m.indptr, None, None, axis)
        g.synchronize(True)
        if use_comm:
            g.set_comm(None)
        outs.append((P, Q))
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


def test_the_product_library_refuses_the_test_transport(monkeypatch):
    """BFH_COMM_TRANSPORT=shm is the knob of libbuffalo_hip_test.so (tests/test_comm_ranks_gpu.py); the product library this process loaded
    says so instead of silently building an RCCL communicator -- and an id made by the test library is refused too."""
    import os
    from buffalo_amd._lib import BuffaloHipError
    from buffalo_amd.backend import Comm
    monkeypatch.setenv("BFH_COMM_TRANSPORT", "shm")
    with pytest.raises(BuffaloHipError, match="TEST transport"):
        Comm.unique_id()
    monkeypatch.delenv("BFH_COMM_TRANSPORT")
    uid = (b"BFHSHM1\x00" + os.urandom(16)).ljust(128, b"\x00")
    with pytest.raises(BuffaloHipError, match="TEST transport"):
        Comm(1, 0, uid, 0)
