"""WARP parity at BASELINE scale (missing #3): the ML-20M-shaped matrix at configs[4]'s d = 256, rows of hundreds to
thousands of positives, the 8-deep pre-sample path (T > 2) and the cached key runs the full-size epoch uses -- against the oracle
on a sample of the users, number for number.

P and Q are frozen INSIDE an epoch (warp.cc:157-159 accumulates gradients; the optimizer step comes with update_parameters), so
the row a user ends the epoch with depends on its own positives only, and the shard offset keeps the sampler's counters global (warp.hip: gpos = nnz_offset + shift + t; oracle: add_jobs job.add(S, beg + nnz_offset_)):
the oracle run on a range of users with the range's global position reproduces exactly what the full epoch did to them
(/root/reference/lib/algo_impl/warp/warp.cc:128-158 is the loop being reproduced: draw, skip seen, count the trial, score,
accept the first violator with the rank weight log((I - |seen| - 1) / trial))."""
import numpy as np
import pytest

import helpers as H
from conftest import warp_opt

pytestmark = pytest.mark.gpu

D = 256


def _warm_model(csr, seed):
    """Factors in the regime training lives in: a user's row leans towards its own items (p_u ~ sum of its q_i), so positives
    outscore random negatives and the trial loop has to search (T >= 3) instead of accepting the first draw."""
    import scipy.sparse as sp
    U, I = csr.num_users, csr.num_items
    rng = np.random.default_rng(seed)
    Q = rng.normal(scale=0.6 / np.sqrt(D), size=(I, D)).astype(np.float32)          # |q| ~ 0.6: inside the unit ball, so that
    # update_parameters' projection (warp.cc:192-201, Q-12) leaves the frozen item side untouched
    A = sp.csr_matrix((np.ones(csr.nnz, np.float32), csr.keys, np.concatenate([[0], csr.indptr])), shape=(U, I))
    deg = np.maximum(np.diff(np.concatenate([[0], csr.indptr])), 1).astype(np.float32)
    P = (A @ Q) / np.sqrt(deg)[:, None]
    assert np.linalg.norm(Q, axis=1).max() < 1.0
    P += rng.normal(scale=0.1 / np.sqrt(D), size=P.shape).astype(np.float32)
    return np.ascontiguousarray(P.astype(np.float32)), Q


def test_trial_counts_and_gradients_at_ml20m_scale(oracle):
    import bench
    from buffalo_amd.backend import CyWARP
    from buffalo_amd.synth import CSR
    csr = bench.load_matrix("ml20m", 7)
    U, I, nnz = csr.num_users, csr.num_items, csr.nnz
    P0, Q = _warm_model(csr, 5)
    Qb = np.zeros((I, 1), np.float32)
    starts = np.concatenate([[0], csr.indptr[:-1]])
    deg = csr.indptr - starts
    # margin: the 30 % quantile of (x_ui - x_uj) over sampled triples -> about 3 of 10 draws violate, T ~ 1 / 0.3 counted trials
    rng = np.random.default_rng(1)
    su = rng.integers(0, U, 200000)
    si = csr.keys[starts[su] + (rng.random(200000) * deg[su]).astype(np.int64)]
    sj = rng.integers(0, I, 200000)
    diff = np.einsum("ij,ij->i", P0[su], Q[si] - Q[sj])
    thr = float(np.quantile(diff, 0.30))
    opt = warp_opt(d=D, lr=0.05, min_lr=0.05, num_iters=1, random_seed=11, max_trials=500, threshold=thr, optimizer="adagrad")
    EPOCHS = 1      # (one epoch: the item side moves at its end -- WARP has no update_i / update_j -- and the sample alone cannot reproduce that)

    # ---- the full-size epochs on the device ----
    Pg, Qg = P0.copy(), Q.copy()
    full = CyWARP()
    assert full.init(H.write_opt(dict(opt, accelerator=True)))
    full.set_mode("warp_presample", 8)                 # the 8-deep pre-sample a run picks by itself once T > 2 (the sample below: 4 deep)
    full.initialize_model(Pg, Qg, Qb.copy(), nnz, True)
    full.set_resident_csr(csr.indptr, csr.keys)
    T_full = []
    for _ in range(EPOCHS):
        full.reset_stats()
        full.add_jobs(0, U, csr.indptr, None)
        full.update_parameters()
        st = full.stats()
        T_full.append(st["scored_negatives"] / nnz)
    full.synchronize(True)
    assert T_full[-1] >= 3.0, T_full                   # the searching regime, 8-deep pre-sample
    del full

    # ---- the sample: the 24 heaviest users one by one + 20 stretches of 100 users, in ascending user order ----
    heavy = np.sort(np.argsort(-deg)[:24])
    ranges = [(int(u), int(u) + 1) for u in heavy] + [(int(a), int(a) + 100) for a in np.linspace(0, U - 100, 20).astype(int)]
    ranges.sort()
    merged = []
    for a, b in ranges:                                 # a heavy user may sit inside a stretch
        if merged and a < merged[-1][1]:
            merged[-1] = (merged[-1][0], max(merged[-1][1], b))
        else:
            merged.append((a, b))
    users = np.concatenate([np.arange(a, b) for a, b in merged])
    sub_deg = deg[users]
    sub_indptr = np.cumsum(sub_deg).astype(np.int64)
    sub_keys = np.concatenate([csr.keys[starts[u]:csr.indptr[u]] for u in users]).astype(np.int32)
    assert sub_deg.max() >= 2000 and len(users) >= 2000, (sub_deg.max(), len(users))
    # per range: rows [a_s, b_s) of the sub-matrix, the sub-matrix position and the GLOBAL position of its first entry
    pieces, row = [], 0
    for a, b in merged:
        n_rows = b - a
        l = 0 if row == 0 else int(sub_indptr[row - 1])
        pieces.append((row, row + n_rows, l, int(sub_indptr[row + n_rows - 1]), int(starts[a])))
        row += n_rows

    def sampled_run(obj, is_hip):
        counts = []
        for _ in range(EPOCHS):
            for (a_s, b_s, l0, l1, g0) in pieces:
                obj.set_shard(g0 - l0, 1)
                obj.add_jobs(a_s, b_s, sub_indptr, np.ascontiguousarray(sub_keys[l0:l1]))
            obj.update_parameters()
            st = obj.stats()
            counts.append((st["scored_negatives"], st["accepted"] if is_hip else st["updates"]))
        return counts

    Po = np.ascontiguousarray(P0[users])
    o = oracle.OracleWARP()
    assert o.init(H.write_opt(dict(opt, accelerator=False, num_workers=1)))
    o.initialize_model(Po, Q.copy(), Qb.copy(), nnz)
    o.set_cumulative_table(np.zeros(I, np.int64), I)
    o.set_modes(sampler="counter", pos_order="csr", inline=True)
    o.launch_workers()
    c_or = sampled_run(o, False)
    o.join()

    Ps = np.ascontiguousarray(P0[users])
    sub = CyWARP()
    assert sub.init(H.write_opt(dict(opt, accelerator=True)))
    sub.initialize_model(Ps, Q.copy(), Qb.copy(), nnz)
    sub.set_cumulative_table(np.zeros(I, np.int64), I)
    sub.set_placeholder(sub_indptr, int(max(l1 - l0 for (_, _, l0, l1, _) in pieces)) + 1)
    sub.initialize_model(Ps, Q.copy(), Qb.copy(), nnz, True)
    sub.set_cumulative_table(np.zeros(I, np.int64), I)
    c_hip = sampled_run(sub, True)
    sub.synchronize(True)
    print("\nWARP at ML-20M scale, d = 256, threshold %.4f: T per epoch (full run) %s; %d sampled users (longest row %d), "
          "(scored negatives, accepted) per epoch: oracle %s  device on the sample %s" % (thr, ["%.2f" % t for t in T_full], len(users), int(sub_deg.max()), c_or, c_hip))
    assert c_hip == c_or, (c_hip, c_or)                 # identical trial sequences and accept decisions (cumulative per epoch)
    e_sub, e_full = H.relerr(Ps, Po), H.relerr(Pg[users], Po)
    print("   P rows of the sample after %d epochs: device on the sample vs oracle %.2e, full-size run vs oracle %.2e" % (EPOCHS, e_sub, e_full))
    assert not np.array_equal(Po, P0[users])
    assert e_sub < 1e-4, e_sub
    assert e_full < 1e-4, e_full


def test_results_at_a_tenth_of_configs4_size(oracle):
    """a results check at the configs[4] SHAPE (bench.py's generator: 100 positives per user, one per band of the catalogue, popularity-skewed;
    d = 256, adagrad, the benchmark's WARP options) at 1 M users x 100 K items / 100 M interactions -- a tenth of configs[4] on each axis, what the driver's suite
    can hold.  The full-size device epoch against the oracle (counter sampler, one inline worker: /root/reference/lib/algo_impl/warp/warp.cc:128-158 statement by
    statement) on the first 100 K users from the same initial state: identical trial and accept counts (against a device run on those users alone -- the full run's
    counters are totals) and the users' accumulated gradient rows to 1e-4 (the rows after adagrad's first step -- lr * sign(g) -- are reported as a share of
    coordinates: a coordinate whose hundred contributions cancel to rounding noise may step the other way).  bench.py's warp_c5 extra / --workload warp_c5 run the same check (warp_results_check)
    at the full 10 M x 1 M size on 20 K users."""
    import bench
    from buffalo_amd.backend import CyWARP
    U, I, d = 1_000_000, 100_000, bench.WARP_D
    indptr, keys, P, Q, Qb = bench.warp_c5_inputs(users=U, items=I)
    nnz = int(keys.shape[0])
    assert nnz == 100 * U and int(keys.max()) < I
    n_chk = 100_000
    P0, Q0 = np.ascontiguousarray(P[:n_chk]).copy(), Q.copy()
    g = CyWARP()
    assert g.init(H.write_opt(bench.WARP_OPT))
    g.sync_every_epoch = False
    g.initialize_model(P, Q, Qb, nnz, True)
    g.set_resident_csr(indptr, keys)
    g.add_jobs(0, U, indptr, None)
    grads = g.device_tensor("gradP", (U, d))[:n_chk].cpu().numpy().copy()      # the accumulated gradient rows, before the optimizer step
    g.update_parameters()
    st = g.stats()
    del g
    rc = bench.warp_results_check(indptr, keys, P0, Q0, np.zeros((I, 1), np.float32), grads, n_chk, nnz, bench.WARP_OPT)
    print("\nWARP at 1 M x 100 K / 100 M nnz, d = 256: full epoch T %.2f accepted %.3f; check %s" % (st["scored_negatives"] / nnz, st["accepted"] / nnz, rc))
    assert (rc["scored_negatives_device"], rc["accepted_device"]) == (rc["scored_negatives_oracle"], rc["accepted_oracle"]), rc
    assert rc["grad_rows_max_abs"] > 1e-3, rc
    assert rc["grad_rows_full_run_vs_oracle"] < 1e-4 and rc["grad_rows_sample_run_vs_oracle"] < 1e-4, rc
    assert rc["rows_after_step_share_of_coordinates_apart"] < 1e-5, rc
    assert rc["agrees_with_device"]
