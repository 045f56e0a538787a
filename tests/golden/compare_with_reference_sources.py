"""The oracle next to the REFERENCE's own SGD sources, same inputs, same calls (TEST INFRASTRUCTURE).

    BUFFALO_REF_SGD_EXACT=1 BUFFALO_ORACLE_LIB=oracle/_ref/libbuffalo_oracle_exact.so python tests/golden/compare_with_reference_sources.py
    python tests/golden/compare_with_reference_sources.py            # both sides built with the reference's own flags

One side: `OracleBPRMF` / `OracleWARP` (oracle/buffalo_oracle.cc, a restatement).  Other side: `RefBPRMF` / `RefWARP`
(oracle/ref_sgd.py): /root/reference/lib/algo.cc, lib/algo_impl/bpr/bpr.cc, warp/warp.cc, lib/misc/log.cc compiled UNMODIFIED against
oracle/stand_in_3rd.  Both are driven like the reference's CPU front drives CyBPRMF / CyWARP -- init from the option file,
initialize_model, set_cumulative_table, launch_workers, epochs of add_jobs (two chunks) + update_parameters, join, compute_loss -- in
the reference's own modes (std::mt19937, unordered_set order, worker thread + queue), with ONE worker and min_lr = lr so that nothing
depends on thread timing (Q-8), and a pause after wait_until_done so that the optimizer pass finds the queue's work finished.
Prints one JSON line per configuration: max |difference| of P, Q, Qb and whether the losses are equal.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

BPR = [dict(), dict(optimizer="adagrad"), dict(optimizer="adam", per_coordinate_normalize=True), dict(optimizer="adam", use_bias=False),
       dict(num_negative_samples=2, use_bias=False), dict(sampling_power=1.0, verify_neg=False), dict(update_i=False),
       dict(update_j=False, reg_u=0.1), dict(optimizer="adagrad", per_coordinate_normalize=True, num_negative_samples=3),
       dict(d=300, num_negative_samples=2), dict(d=1), dict(update_i=False, update_j=False), dict(reg_u=0.0, reg_i=0.0, reg_j=0.0, reg_b=0.0),
       dict(sampling_power=0.5), dict(random_seed=777, optimizer="adam", sampling_power=2.0)]
WARP = [dict(max_trials=10, threshold=0.3), dict(max_trials=20, threshold=0.5, score_func="l2"),
        dict(max_trials=8, threshold=0.2, optimizer="adam", per_coordinate_normalize=True),
        dict(max_trials=30, threshold=0.5, reg_u=0.01, reg_i=0.02, reg_j=0.03), dict(max_trials=500, threshold=1.0, optimizer="adagrad"),
        dict(max_trials=2, threshold=0.5), dict(max_trials=3, threshold=100.0), dict(d=256, max_trials=12, threshold=0.4, score_func="l2", reg_i=0.02)]


ALS = [  # (options, chunks): the iALS++ path is fed in one chunk -- for start_x > 0 stock buffalo sizes its Yui buffer by the wrong
        # offset (als.cc:251, Q-15) and overruns it
    (dict(optimizer="llt"), 2), (dict(optimizer="ldlt", adaptive_reg=True), 2), (dict(optimizer="manual_cg"), 2),
    (dict(optimizer="manual_cg", compute_loss_on_training=False, alpha=3.0), 3), (dict(optimizer="ialspp", d=24, block_size=7), 1),
    (dict(optimizer="ialspp", d=48, block_size=32, num_workers=2), 1), (dict(d=128, block_size=32, num_workers=2), 1),
]


def als_rows():
    """ALS: the reference's als.cc on the stand-ins against OracleALS.  Dense products and Cholesky solves are each side's own loops, so
    this is a tolerance comparison: two epochs from the same factors, the largest and the median per-row difference, the loss pairs."""
    import helpers as H
    from conftest import als_opt, tiny_csr
    from oracle import oracle, ref_sgd
    csr = tiny_csr(U=120, I=90, density=0.2, seed=5, counts=True)
    t = csr.transpose()

    def run(cls, opt, P, Q, chunks):
        o = cls()
        path = H.write_opt(opt)
        assert o.init(path)
        os.unlink(path)
        o.initialize_model(P, Q)
        losses = []
        for _ in range(2):
            for axis, m in ((0, csr), (1, t)):
                o.precompute(axis)
                n = w = 0.0
                for a, b in H.chunks_of(m, chunks):
                    keys, vals = H.chunk_arrays(m, a, b)
                    x, y = o.partial_update(a, b, m.indptr, keys, vals, axis)
                    n, w = n + x, w + y
                losses.append((n, w))
        return losses
    for kw, chunks in ALS:
        d = kw.get("d", 20)
        opt = als_opt(**dict(dict(d=d, num_workers=1, reg_u=0.1, reg_i=0.2), **kw))
        rng = np.random.default_rng(2)
        P0 = np.abs(rng.normal(scale=0.1, size=(csr.num_users, d))).astype(np.float32)
        Q0 = np.abs(rng.normal(scale=0.1, size=(csr.num_items, d))).astype(np.float32)
        A, B = [P0.copy(), Q0.copy()], [P0.copy(), Q0.copy()]
        la, lb = run(oracle.OracleALS, opt, *A, chunks), run(ref_sgd.RefALS, opt, *B, chunks)
        per_row = np.concatenate([np.abs(a - b).max(axis=1) / np.abs(a).max() for a, b in zip(A, B)])
        print(json.dumps({"algo": "als", "options": kw, "max_rel_diff": float(per_row.max()), "median_rel_diff": float(np.median(per_row)),
                          "loss_rel_diff": max(abs(x - y) / max(1.0, abs(x)) for u, v in zip(la, lb) for x, y in zip(u, v)),
                          "moved": float(np.abs(A[0] - P0).max())}), flush=True)


def eals_cfr_rows():
    """eALS (eals.cc / eals.hpp; its Gramians through a written-out ssyrk) and CFR (cfr.cc) on the stand-ins against OracleEALS / OracleCFR:
    two epochs in the order the fronts call them, largest relative difference per array, the loss values."""
    import helpers as H
    from conftest import tiny_csr
    from oracle import oracle, ref_sgd
    for shape, d in (((120, 70, 0.12, 1), 20), ((120, 70, 0.12, 1), 128), ((600, 60, 0.9, 2), 40)):
        csr = tiny_csr(U=shape[0], I=shape[1], density=shape[2], seed=shape[3], counts=True)
        t = csr.transpose()
        opt = {"d": d, "num_workers": 2, "alpha": 2.0, "reg_u": 0.1, "reg_i": 0.2, "num_iters": 3, "c0": 0.5, "exponent": 0.5, "model_path": "", "data_opt": {}}
        rng = np.random.default_rng(0)
        P0 = rng.normal(scale=0.3, size=(csr.num_users, d)).astype(np.float32)
        Q0 = rng.normal(scale=0.3, size=(csr.num_items, d)).astype(np.float32)
        pop = np.diff(np.concatenate([[0], t.indptr])).astype(np.float32)
        pop /= pop.max()
        Cw = (0.5 * pop ** 0.5 / (pop ** 0.5).sum()).astype(np.float32)
        res = []
        for cls in (oracle.OracleEALS, ref_sgd.RefEALS):
            P, Q = P0.copy(), Q0.copy()
            o = cls()
            assert o.init(H.write_opt(opt))
            o.initialize_model(P, Q, Cw)
            o.precompute_cache(csr.nnz, csr.indptr, csr.keys, 0)
            o.precompute_cache(csr.nnz, t.indptr, t.keys, 1)
            ls = []
            for _ in range(2):
                for axis, m in ((0, csr), (1, t)):
                    assert o.update(m.indptr, m.keys, m.vals, axis)
                    ls += list(o.estimate_loss(csr.nnz, m.indptr, m.keys, m.vals, axis))
            res.append(({"P": P, "Q": Q}, ls))
        _emit("eals", {"shape": shape[:2], "d": d}, res, {"P": P0})
    Uu, Ii = 150, 90
    csr = tiny_csr(U=Uu, I=Ii, density=0.1, seed=3, counts=True)
    t = csr.transpose()
    ctx = tiny_csr(U=Ii, I=Ii, density=0.15, seed=4, counts=True)
    for kw in (dict(optimizer="llt"), dict(optimizer="manual_cg"), dict(optimizer="ldlt", l=1.0, d=40), dict(optimizer="llt", compute_loss=False)):
        opt = {"d": 20, "num_workers": 2, "num_cg_max_iters": 3, "alpha": 4.0, "l": 0.7, "eps": 1e-10, "reg_u": 0.1, "reg_i": 0.2, "reg_c": 0.3,
               "compute_loss": True, "optimizer": "llt", "cg_tolerance": 1e-10, "num_iters": 2, "model_path": "", "data_opt": {}}
        opt.update(kw)
        d = opt["d"]

        def arrays():
            rng = np.random.default_rng(9)
            return {n: rng.normal(scale=0.2, size=(r, c)).astype(np.float32)
                    for n, r, c in (("user", Uu, d), ("item", Ii, d), ("context", Ii, d), ("item_bias", Ii, 1), ("context_bias", Ii, 1))}
        res = []
        for cls in (oracle.OracleCFR, ref_sgd.RefCFR):
            A = arrays()
            o = cls()
            assert o.init(H.write_opt(opt))
            for n in ("user", "item", "context", "item_bias", "context_bias"):
                o.set_embedding(A[n], n)
            ls = []
            for _ in range(2):
                o.precompute("item")
                ls.append(o.partial_update_user(0, Uu, csr.indptr, csr.keys, csr.vals))
                o.precompute("user")
                ls.append(o.partial_update_item(0, Ii, t.indptr, t.keys, t.vals, ctx.indptr, ctx.keys, ctx.vals))
                ls.append(o.partial_update_context(0, Ii, ctx.indptr, ctx.keys, ctx.vals))
            res.append((A, ls))
        _emit("cfr", kw, res, arrays())


def _emit(algo, options, res, start):
    (A, la), (B, lb) = res
    rel = {n: float(np.abs(A[n] - B[n]).max() / max(np.abs(A[n]).max(), 1e-30)) for n in A}
    first = next(iter(start))
    print(json.dumps({"algo": algo, "options": options, "rel_diff": rel, "max_rel_diff": max(rel.values()),
                      "loss_rel_diff": max([abs(x - y) / max(1.0, abs(x)) for x, y in zip(la, lb)] or [0.0]),
                      "moved": float(np.abs(A[first] - start[first]).max())}), flush=True)


def main():
    if "als" in sys.argv[1:]:
        return als_rows()
    if "eals_cfr" in sys.argv[1:]:
        return eals_cfr_rows()
    import helpers as H
    from conftest import bpr_opt, tiny_csr, warp_opt
    from oracle import oracle, ref_sgd
    csr = tiny_csr(U=48, I=70, density=0.18, seed=13)
    epochs = 3

    def run(cls, opt, P, Q, Qb):
        o = cls()
        path = H.write_opt(opt)
        assert o.init(path)
        os.unlink(path)
        o.initialize_model(P, Q, Qb, csr.nnz)
        o.set_cumulative_table(H.cum_table(csr, opt), csr.num_items)
        o.launch_workers()
        for _ in range(epochs):
            for a, b in H.chunks_of(csr, 2):
                keys, _ = H.chunk_arrays(csr, a, b)
                o.add_jobs(a, b, csr.indptr, keys)
            o.wait_until_done()            # "queue empty" only (algo.cc:467-472): give the single worker time to finish its last job
            time.sleep(0.05)
            o.update_parameters()
        o.join()
        users, pos, neg = (np.arange(12, dtype=np.int32), np.arange(12, dtype=np.int32), np.arange(20, 32, dtype=np.int32))
        return o.compute_loss(users, pos, neg)

    def run_decay(cls, opt, P, Q, Qb):
        """The learning-rate thread (algo.cc:261-306): with min_lr < lr a job carries the rate add_jobs reads when it queues it, which the
        progress thread lowers as jobs complete -- timing, in general (Q-8).  One user per add_jobs call, each call made only after the
        previous job is done and accounted for, makes every job's rate a function of completed work alone."""
        o = cls()
        path = H.write_opt(opt)
        assert o.init(path)
        os.unlink(path)
        o.initialize_model(P, Q, Qb, csr.nnz)
        o.set_cumulative_table(H.cum_table(csr, opt), csr.num_items)
        o.launch_workers()
        for _ in range(epochs):
            for x in range(csr.num_users):
                beg = int(csr.indptr[x - 1]) if x else 0
                o.add_jobs(x, x + 1, csr.indptr, np.ascontiguousarray(csr.keys[beg:int(csr.indptr[x])]))
                time.sleep(0.010)          # one user's job takes microseconds; then the queue is empty and wait_until_done returns at once
                o.wait_until_done()        # (called first it would find the job still queued and sleep its 100 ms, algo.cc:470)
            o.update_parameters()
        o.join()
        return 0.0

    for kw in (dict(), dict(use_bias=False, num_negative_samples=2)):
        opt = bpr_opt(d=20, lr=0.05, min_lr=0.004, num_iters=epochs, random_seed=7, num_workers=1, **kw)
        rng = np.random.default_rng(1)
        P0 = rng.normal(scale=0.3, size=(csr.num_users, 20)).astype(np.float32)
        Q0 = rng.normal(scale=0.3, size=(csr.num_items, 20)).astype(np.float32)
        Qb0 = rng.normal(scale=0.1, size=(csr.num_items, 1)).astype(np.float32) * (1 if opt["use_bias"] else 0)
        A, B = [x.copy() for x in (P0, Q0, Qb0)], [x.copy() for x in (P0, Q0, Qb0)]
        run_decay(oracle.OracleBPRMF, opt, *A)
        run_decay(ref_sgd.RefBPRMF, opt, *B)
        const = [x.copy() for x in (P0, Q0, Qb0)]                 # the same run at a constant rate: the decay must have made a difference
        run_decay(oracle.OracleBPRMF, dict(opt, min_lr=opt["lr"]), *const)
        print(json.dumps({"algo": "bpr", "options": dict(kw, lr_decay=True), "max_abs_diff": [float(np.abs(a - b).max()) for a, b in zip(A, B)],
                          "identical": [bool(np.array_equal(a, b)) for a, b in zip(A, B)], "loss": [0.0, 0.0],
                          "moved": float(np.abs(A[0] - P0).max()), "scale": float(np.abs(A[0]).max()),
                          "decay_effect": float(np.abs(A[0] - const[0]).max())}), flush=True)

    for algo, cases, ocls, rcls, mk, d in (("bpr", BPR, oracle.OracleBPRMF, ref_sgd.RefBPRMF, bpr_opt, 20),
                                           ("warp", WARP, oracle.OracleWARP, ref_sgd.RefWARP, warp_opt, 24)):
        for kw in cases:
            opt = mk(**dict(dict(d=d, lr=0.05, min_lr=0.05, num_iters=epochs, random_seed=7, num_workers=1), **kw))
            d = opt["d"]
            rng = np.random.default_rng(1)
            P0 = rng.normal(scale=0.3, size=(csr.num_users, d)).astype(np.float32)
            Q0 = rng.normal(scale=0.3, size=(csr.num_items, d)).astype(np.float32)
            Qb0 = rng.normal(scale=0.1, size=(csr.num_items, 1)).astype(np.float32)
            if algo == "warp" or not opt.get("use_bias", False):
                Qb0 *= 0
            A, B = [x.copy() for x in (P0, Q0, Qb0)], [x.copy() for x in (P0, Q0, Qb0)]
            la, lb = run(ocls, opt, *A), run(rcls, opt, *B)
            print(json.dumps({"algo": algo, "options": kw, "max_abs_diff": [float(np.abs(a - b).max()) for a, b in zip(A, B)],
                              "identical": [bool(np.array_equal(a, b)) for a, b in zip(A, B)], "loss": [la, lb],
                              "moved": float(np.abs(A[0] - P0).max()), "scale": float(np.abs(A[0]).max())}), flush=True)


if __name__ == "__main__":
    main()
