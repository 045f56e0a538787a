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
       dict(update_j=False, reg_u=0.1), dict(optimizer="adagrad", per_coordinate_normalize=True, num_negative_samples=3)]
WARP = [dict(max_trials=10, threshold=0.3), dict(max_trials=20, threshold=0.5, score_func="l2"),
        dict(max_trials=8, threshold=0.2, optimizer="adam", per_coordinate_normalize=True),
        dict(max_trials=30, threshold=0.5, reg_u=0.01, reg_i=0.02, reg_j=0.03), dict(max_trials=500, threshold=1.0, optimizer="adagrad")]


def main():
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
            time.sleep(0.15)
            o.update_parameters()
        o.join()
        users, pos, neg = (np.arange(12, dtype=np.int32), np.arange(12, dtype=np.int32), np.arange(20, 32, dtype=np.int32))
        return o.compute_loss(users, pos, neg)

    for algo, cases, ocls, rcls, mk, d in (("bpr", BPR, oracle.OracleBPRMF, ref_sgd.RefBPRMF, bpr_opt, 20),
                                           ("warp", WARP, oracle.OracleWARP, ref_sgd.RefWARP, warp_opt, 24)):
        for kw in cases:
            opt = mk(d=d, lr=0.05, min_lr=0.05, num_iters=epochs, random_seed=7, num_workers=1, **kw)
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
