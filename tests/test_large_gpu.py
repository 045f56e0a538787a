"""Factor matrices above 4 GiB (4.3 M users x 256 floats = 4.4 GB): every row address on the path has to be
64-bit.  The rows that sit beyond the 4 GiB mark are checked against the oracle run on just those rows:

* BPRMF with the item side frozen (update_i = update_j = False): a user's row then depends on its own triples
  only, and the shard offset keeps the sampler's counters global, so the oracle on the last users reproduces
  exactly what the full run did to them;
* WARP (adagrad) with the item side frozen: the same argument for the gradient / velocity rows;
* ALS (vdim 256, the wide kernel): a row update depends on the other side and FF only -- user rows beyond the mark
  vs the oracle on a sub-problem; item rows gather q rows from the 4.4 GB matrix (64-bit gather offsets).

Set BFH_SKIP_LARGE=1 to skip (about a minute, 15 GB of host memory)."""
import os

import numpy as np
import pytest

import helpers as H
from conftest import als_opt, bpr_opt, warp_opt

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("BFH_SKIP_LARGE") == "1", reason="BFH_SKIP_LARGE=1")]

U, I, D, DEG = 4_300_000, 50_000, 256, 8
TAIL = 48          # users compared against the oracle (their rows start beyond 4 GiB)


def _problem():
    from buffalo_amd.synth import CSR
    step = I // DEG
    u = np.arange(U, dtype=np.int64)
    keys = ((u * 7) % step)[:, None] + (np.arange(DEG, dtype=np.int64) * step)[None, :]      # sorted, distinct per row
    indptr = (u + 1) * DEG
    csr = CSR(U, I, indptr, np.ascontiguousarray(keys.reshape(-1).astype(np.int32)), np.ones(U * DEG, np.float32))
    rng = np.random.default_rng(0)
    base = np.abs(rng.normal(scale=0.05, size=(65536, D))).astype(np.float32)
    P = np.ascontiguousarray(np.tile(base, (U // 65536 + 1, 1))[:U])
    P += (np.arange(U, dtype=np.float32) % 977)[:, None] * 1e-5          # rows differ
    Q = np.abs(rng.normal(scale=0.05, size=(I, D))).astype(np.float32)
    assert P.nbytes > (1 << 32)
    return csr, P, Q


def test_rows_beyond_4gib_match_the_oracle(oracle):
    from buffalo_amd.backend import CyALS, CyBPR
    from buffalo_amd.synth import CSR
    csr, P, Q = _problem()
    u0 = U - TAIL
    sub = CSR(TAIL, I, csr.indptr[u0:] - u0 * DEG, np.ascontiguousarray(csr.keys[u0 * DEG:]), np.ones(TAIL * DEG, np.float32))

    # ---------------- BPRMF, item side frozen ----------------
    opt = bpr_opt(d=D, lr=0.05, min_lr=0.05, num_iters=1, update_i=False, update_j=False, use_bias=False, random_seed=11,
                  reg_u=0.01, accelerator=True)
    Qb = np.zeros((I, 1), np.float32)
    Pg, Qg = P.copy(), Q.copy()
    obj = CyBPR()
    assert obj.init(H.write_opt(opt))
    obj.set_mode("hogwild_atomic", 1)      # user-major walk: a user's triples are applied in CSR order, like the oracle's
    obj.initialize_model(Pg, Qg, Qb, csr.nnz, True)
    obj.set_cumulative_table(np.zeros(I, np.int64), I)
    obj.set_resident_csr(csr.indptr, csr.keys)
    obj.add_jobs(0, U, csr.indptr, None)
    obj.update_parameters()
    assert np.array_equal(Qg, Q)                                        # frozen item side
    moved = np.abs(Pg[::100003] - P[::100003]).max(axis=1)
    assert np.all(moved > 0) and np.isfinite(Pg[-TAIL:]).all()
    Po, Qo = P[u0:].copy(), Q.copy()
    o = oracle.OracleBPRMF()
    assert o.init(H.write_opt(dict(opt, accelerator=False, num_workers=1)))
    o.initialize_model(Po, Qo, Qb.copy(), csr.nnz)
    o.set_cumulative_table(np.zeros(I, np.int64), I)
    o.set_modes(sampler="counter", pos_order="csr", inline=True)
    o.set_shard(u0 * DEG, 1)                                            # global nnz position of the first compared triple
    o.launch_workers()
    o.add_jobs(0, TAIL, sub.indptr, sub.keys)
    o.update_parameters()
    assert H.relerr(Pg[u0:], Po) < 1e-5
    del obj, Pg, Qg
    # the default item-major walk applies a user's 8 steps in item order at different times: same rows, same
    # addresses, a different (legal Hogwild) order -- the rows beyond the mark agree to the order of lr^2
    Pg, Qg = P.copy(), Q.copy()
    obj = CyBPR()
    assert obj.init(H.write_opt(opt))
    obj.initialize_model(Pg, Qg, Qb, csr.nnz, True)
    obj.set_cumulative_table(np.zeros(I, np.int64), I)
    obj.set_resident_csr(csr.indptr, csr.keys)
    obj.add_jobs(0, U, csr.indptr, None)
    obj.update_parameters()
    assert obj.stats()["merges"] >= 1 and np.array_equal(Qg, Q)
    moved = np.abs(Pg[::100003] - P[::100003]).max(axis=1)
    assert np.all(moved > 0) and np.isfinite(Pg[-TAIL:]).all()
    assert H.relerr(Pg[u0:], Po) < 5e-3, H.relerr(Pg[u0:], Po)
    del obj, Pg, Qg

    # ---------------- WARP, item side frozen: gradP / velocity rows beyond the mark ----------------
    from buffalo_amd.backend import CyWARP
    wopt = warp_opt(d=D, lr=0.05, min_lr=0.05, num_iters=1, update_i=False, update_j=False, random_seed=11, max_trials=20, threshold=0.5,
                    optimizer="adagrad", accelerator=True)
    Pw, Qw = P.copy(), Q.copy()
    Pw[u0:] *= -1.0                                                     # some negative scores so that violators exist
    Pw0 = Pw[u0:].copy()
    warp = CyWARP()
    assert warp.init(H.write_opt(wopt))
    warp.initialize_model(Pw, Qw, np.zeros((I, 1), np.float32), csr.nnz, True)
    warp.set_resident_csr(csr.indptr, csr.keys)
    warp.add_jobs(0, U, csr.indptr, None)
    warp.update_parameters()
    Po, Qo = Pw0.copy(), Q.copy()
    ow = oracle.OracleWARP()
    assert ow.init(H.write_opt(dict(wopt, accelerator=False, num_workers=1)))
    ow.initialize_model(Po, Qo, np.zeros((I, 1), np.float32), csr.nnz)
    ow.set_cumulative_table(np.zeros(I, np.int64), I)
    ow.set_modes(sampler="counter", pos_order="csr", inline=True)
    ow.set_shard(u0 * DEG, 1)
    ow.launch_workers()
    ow.add_jobs(0, TAIL, sub.indptr, sub.keys)
    ow.update_parameters()
    assert not np.array_equal(Pw[u0:], Pw0) and H.relerr(Pw[u0:], Po) < 1e-4
    del warp, Pw, Qw

    # ---------------- ALS vdim 256 (wide kernel) ----------------
    aopt = als_opt(d=D, alpha=4.0, reg_u=0.1, reg_i=0.1, compute_loss_on_training=False, accelerator=True)
    Pa, Qa = P.copy(), Q.copy()
    t = csr.transpose()
    als = CyALS()
    assert als.init(H.write_opt(aopt))
    als.initialize_model(Pa, Qa)
    als.set_resident_csr(0, csr.indptr, csr.keys, csr.vals)
    als.set_resident_csr(1, t.indptr, t.keys, t.vals)
    als.set_mode("als_writeback", 0)
    als.precompute(0)
    als.partial_update(0, U, csr.indptr, None, None, 0)
    als.synchronize(True)
    Po, Qo = P[u0:].copy(), Q.copy()
    oa = oracle.OracleALS()
    assert oa.init(H.write_opt(dict(aopt, accelerator=False, num_workers=8)))
    oa.initialize_model(Po, Qo)
    oa.precompute(0)
    oa.partial_update(0, TAIL, sub.indptr, sub.keys, sub.vals, 0)
    # three fp32 CG steps per block at d = 256 on 8-nnz rows: conditioning-limited (tests/test_als_gpu.py holds the
    # kernels to a float64 envelope); a wrong address shows up as an O(1) error
    assert H.relerr(Pa[u0:], Po) < 2e-2, H.relerr(Pa[u0:], Po)
    assert np.isfinite(Pa[::100003]).all() and not np.array_equal(Pa[::100003], P[::100003])
    # item side: every item row gathers ~690 user rows spread over the whole 4.4 GB matrix
    als.precompute(1)
    als.partial_update(0, I, t.indptr, None, None, 1)
    als.synchronize(True)
    n_items = 6
    Pfull, Qsub = Pa.copy(), Q.copy()
    ob = oracle.OracleALS()
    assert ob.init(H.write_opt(dict(aopt, accelerator=False, num_workers=64)))
    ob.initialize_model(Pfull, Qsub)
    ob.precompute(1)
    keys, vals = H.chunk_arrays(t, 0, n_items)
    ob.partial_update(0, n_items, t.indptr, keys, vals, 1)
    assert H.relerr(Qa[:n_items], Qsub[:n_items]) < 2e-2, H.relerr(Qa[:n_items], Qsub[:n_items])
