"""Secondary measurements for DESIGN.md / profiles (not the driver's bench contract):
ALS (config #3), WARP (config #5 scaled to one GPU), BPR adagrad, and the PCIe-inclusive BPR rate
when the boundary hands over host keys every epoch like the reference does."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import bpr_options, load_matrix, write_opt  # noqa: E402
from buffalo_amd import synth  # noqa: E402
from buffalo_amd.backend import CyALS, CyBPR, CyWARP  # noqa: E402

out = {}
csr = load_matrix("ml20m", 7)
U, I, nnz = csr.num_users, csr.num_items, csr.nnz
which = sys.argv[1:] or ["als", "warp", "bpr_adagrad", "bpr_pcie"]

if "als" in which:
    rng = np.random.default_rng(7)
    vals = (1 + rng.poisson(1.0, size=nnz)).astype(np.float32)
    c2 = synth.CSR(U, I, csr.indptr, csr.keys, vals)
    t = c2.transpose()
    for d, with_loss in ((128, False), (128, True), (32, False), (32, True), (256, False)):
        P, Q, _ = synth.init_factors(U, I, d, seed=7)
        opt = {"evaluation_on_learning": False, "compute_loss_on_training": with_loss, "early_stopping_rounds": 0, "save_best": False,
               "evaluation_period": 1, "save_period": 10, "random_seed": 7, "validation": {}, "adaptive_reg": False,
               "save_factors": False, "accelerator": True, "d": d, "num_iters": 10, "num_workers": 8, "hyper_threads": 256,
               "num_cg_max_iters": 3, "reg_u": 0.1, "reg_i": 0.1, "alpha": 8.0, "optimizer": "manual_cg", "cg_tolerance": 1e-10,
               "block_size": 32, "eps": 1e-10, "model_path": "", "data_opt": {}}
        g = CyALS()
        assert g.init(write_opt(opt))
        g.initialize_model(P, Q)
        g.set_resident_csr(0, c2.indptr, c2.keys, c2.vals)
        g.set_resident_csr(1, t.indptr, t.keys, t.vals)
        g.set_mode("als_writeback", 0)
        if os.environ.get("ALS_DEBUG"):
            g.set_mode("als_debug", int(os.environ["ALS_DEBUG"]))
        if os.environ.get("ALS_INREG"):
            g.set_mode("als_inreg", int(os.environ["ALS_INREG"]))

        def epoch():
            for axis, mat in ((0, c2), (1, t)):
                g.precompute(axis)
                g.partial_update(0, mat.num_users, mat.indptr, None, None, axis)
        epoch()
        g.reset_stats()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            epoch()
        dt = (time.perf_counter() - t0) / n
        st = g.stats()
        name = "als_d%d%s" % (d, "_loss" if with_loss else "")
        T_ = d // 32

        tiles = T_ * (T_ + 1) // 2
        mfma_flop = 2 * nnz * tiles * 2 * 32 * 32      # issued: upper-triangle tiles, both half-epochs
        out[name] = {"epoch_ms": dt * 1e3, "mfma_issued_TFLOPs": mfma_flop / (st["kernel_ms"] / n * 1e-3) / 1e12, "kernel_ms_per_epoch": st["kernel_ms"] / n, "gramian_ms_per_epoch": st["aux_ms"] / n,
                              "interactions_per_s": 2 * nnz / dt, "optimizer": "ialspp(bs=32)" if d >= 128 else "manual_cg(3)"}
        print("als", name, out[name], flush=True)

if "warp" in which:
    d = 256
    P, Q, Qb = synth.init_factors(U, I, d, seed=7, signed=True)
    Qb *= 0
    opt = {"evaluation_on_learning": False, "compute_loss_on_training": False, "early_stopping_rounds": 0, "save_best": False,
           "evaluation_period": 5, "save_period": 10, "random_seed": 7, "validation": {}, "accelerator": True, "num_workers": 8,
           "hyper_threads": 256, "num_iters": 10, "d": d, "threshold": 1.0, "score_func": "dot", "max_trials": 500, "update_i": True,
           "update_j": True, "reg_u": 0.0, "reg_i": 0.0, "reg_j": 0.0, "optimizer": "adagrad", "lr": 0.05, "min_lr": 0.0001,
           "beta1": 0.9, "beta2": 0.999, "eps": 1e-10, "per_coordinate_normalize": False, "model_path": "", "data_opt": {}}
    g = CyWARP()
    assert g.init(write_opt(opt))
    g.sync_every_epoch = False
    g.initialize_model(P, Q, Qb, nnz, True)
    g.set_resident_csr(csr.indptr, csr.keys)
    ep = []
    for e in range(4):
        g.reset_stats()
        t0 = time.perf_counter()
        g.add_jobs(0, U, csr.indptr, None)
        t1 = time.perf_counter()
        g.update_parameters()
        t2 = time.perf_counter()
        st = g.stats()
        T = st["scored_negatives"] / nnz
        alg = st["accepted"] * ((8 + st["scored_negatives"] / max(st["accepted"], 1)) * 4 * d + 4) \
            + (nnz - st["accepted"]) * (2 * 4 * d + 4)   # rough: SURVEY 8(d) formula with measured T
        ep.append({"epoch": e, "trial_kernel_ms": st["kernel_ms"], "optimizer_ms": st["optimizer_ms"], "wall_ms": (t2 - t0) * 1e3,
                   "positives_per_s": nnz / (t1 - t0), "mean_scored_negatives_T": T, "accepted_frac": st["accepted"] / nnz,
                   "algorithmic_GBps": alg / (st["kernel_ms"] * 1e-3) / 1e9})
        print("warp", ep[-1], flush=True)
    out["warp_ml20m_d256"] = ep

if "bpr_adagrad" in which:
    P, Q, Qb = synth.init_factors(U, I, 128, seed=7)
    g = CyBPR()
    assert g.init(write_opt(bpr_options(10, optimizer="adagrad", lr=0.05)))
    g.sync_every_epoch = False
    g.initialize_model(P, Q, Qb, nnz, True)
    g.set_cumulative_table(np.zeros(I, np.int64), I)
    g.set_resident_csr(csr.indptr, csr.keys)
    g.add_jobs(0, U, csr.indptr, None)
    g.update_parameters()
    g.reset_stats()
    t0 = time.perf_counter()
    for _ in range(3):
        g.add_jobs(0, U, csr.indptr, None)
        g.update_parameters()
    dt = (time.perf_counter() - t0) / 3
    st = g.stats()
    out["bpr_adagrad"] = {"epoch_ms": dt * 1e3, "accumulate_kernel_ms": st["kernel_ms"] / 3, "optimizer_ms": st["optimizer_ms"] / 3,
                          "optimizer_GBps": (U + I) * 128 * 4 * 6 / (st["optimizer_ms"] / 3 * 1e-3) / 1e9}
    print("bpr_adagrad", out["bpr_adagrad"], flush=True)

if "bpr_pcie" in which:
    # drop-in call pattern of the reference: host keys handed over every epoch + P,Q,Qb copied back.  pin_host = 0 (default since round 5): the
    # copy-back goes through the library's own pinned ring (HostStager); pin_host = 1: the caller's arrays are registered (the default until round 4)
    for pin in (0, 1):
        P, Q, Qb = synth.init_factors(U, I, 128, seed=7)
        g = CyBPR()
        assert g.init(write_opt(bpr_options(10)))
        g.set_mode("pin_host", pin)
        g.initialize_model(P, Q, Qb, nnz)
        g.set_cumulative_table(np.zeros(I, np.int64), I)
        g.set_placeholder(csr.indptr, nnz + 1)
        g.initialize_model(P, Q, Qb, nnz, True)
        g.add_jobs(0, U, csr.indptr, csr.keys)
        g.update_parameters()
        t0 = time.perf_counter()
        for _ in range(3):
            g.add_jobs(0, U, csr.indptr, csr.keys)
            g.update_parameters()          # device optimizer step + D2H of P,Q,Qb (sync_every_epoch=True)
        dt = (time.perf_counter() - t0) / 3
        t0 = time.perf_counter()
        for _ in range(3):
            g.synchronize(True)
        ds = (time.perf_counter() - t0) / 3
        out["bpr_host_buffers_every_epoch_pin_host_%d" % pin] = {"epoch_ms": dt * 1e3, "updates_per_s": nnz / dt, "copy_back_ms": ds * 1e3,
                                                                  "copy_back_GBps": (U + I) * 128 * 4 / ds / 1e9,
                                                                  "note": "80 MB keys H2D (pageable numpy) + fill_rows + kernel + 85 MB D2H per epoch"}
        print("bpr_pcie pin_host=%d" % pin, out["bpr_host_buffers_every_epoch_pin_host_%d" % pin], flush=True)
        del g

if "als_pcie" in which:
    # the reference's call pattern for ALS: keys / vals handed over on every partial_update, updated rows written back every call
    rng = np.random.default_rng(7)
    vals = (1 + rng.poisson(1.0, size=nnz)).astype(np.float32)
    from buffalo_amd import ingest
    col = ingest.coo_to_csr(csr.keys, csr.rows(), vals, I, U)
    from bench import ALS_OPT
    for auto in (1, 0):
        P, Q, _ = synth.init_factors(U, I, 128, seed=7)
        g = CyALS()
        assert g.init(write_opt(ALS_OPT))
        g.set_mode("auto_resident", auto)
        g.initialize_model(P, Q)
        g.set_placeholder(csr.indptr, col["indptr"], nnz + 1)

        def epoch():
            g.precompute(0)
            g.partial_update(0, U, csr.indptr, csr.keys, vals, 0)
            g.precompute(1)
            g.partial_update(0, I, col["indptr"], col["key"], col["val"], 1)
        epoch()
        t0 = time.perf_counter()
        for _ in range(3):
            epoch()
        dt = (time.perf_counter() - t0) / 3
        out["als_host_buffers_auto_resident_%d" % auto] = {"epoch_ms": dt * 1e3, "interactions_per_s": 2 * nnz / dt,
                                                           "note": "keys + vals handed over every call, updated rows copied back every call (als.cu:403)"}
        print("als_pcie auto_resident=%d" % auto, out["als_host_buffers_auto_resident_%d" % auto], flush=True)

if "eals" in which:
    from buffalo_amd.backend import CyEALS
    rng = np.random.default_rng(7)
    vals = (1 + rng.poisson(1.0, size=nnz)).astype(np.float32)
    c2 = synth.CSR(U, I, csr.indptr, csr.keys, vals)
    t = c2.transpose()
    pop = np.bincount(csr.keys, minlength=I).astype(np.float64) ** 0.5
    Cw = (1.0 * pop / pop.sum()).astype(np.float32)
    for d in (32, 128):
        P = rng.normal(scale=0.1, size=(U, d)).astype(np.float32)
        Q = rng.normal(scale=0.1, size=(I, d)).astype(np.float32)
        g = CyEALS()
        assert g.init(write_opt({"d": d, "num_workers": 8, "alpha": 8.0, "reg_u": 0.1, "reg_i": 0.1, "num_iters": 3}))
        g.initialize_model(P, Q, Cw)
        t0 = time.perf_counter()
        g.precompute_cache(nnz, c2.indptr, c2.keys, 0)
        g.precompute_cache(nnz, t.indptr, t.keys, 1)
        cache_s = time.perf_counter() - t0
        l0 = g.estimate_loss(nnz, c2.indptr, c2.keys, c2.vals, 0)
        g.update(c2.indptr, c2.keys, c2.vals, 0), g.update(t.indptr, t.keys, t.vals, 1)
        g.reset_stats()
        t0 = time.perf_counter()
        n = 2
        for _ in range(n):
            g.update(c2.indptr, c2.keys, c2.vals, 0)
            g.update(t.indptr, t.keys, t.vals, 1)
        dt = (time.perf_counter() - t0) / n
        st = g.stats()
        l1 = g.estimate_loss(nnz, c2.indptr, c2.keys, c2.vals, 0)
        out["eals_d%d" % d] = {"epoch_ms": dt * 1e3, "kernel_ms_per_epoch": st["kernel_ms"] / n, "precompute_cache_s_incl_host_sort": cache_s,
                               "interactions_per_s": 2 * nnz / dt, "loss_before": l0[1], "loss_after_3_epochs": l1[1]}
        print("eals", d, out["eals_d%d" % d], flush=True)

if "warp_c5" in which:
    # BASELINE config #5's shape on ONE GPU: 10 M users x 1 M items, 1 B interactions, d=256 (the config shards users over 8)
    U5, I5, deg, d = 10_000_000, 1_000_000, 100, 256
    step = I5 // deg
    t0 = time.perf_counter()
    u = np.arange(U5, dtype=np.int64)
    keys5 = np.ascontiguousarray((((u * 7919) % step)[:, None] + (np.arange(deg, dtype=np.int64) * step)[None, :]).astype(np.int32).reshape(-1))
    indptr5 = (u + 1) * deg
    del u
    rng = np.random.default_rng(7)
    base = (rng.normal(size=(65536, d)) / d).astype(np.float32)
    P5 = np.ascontiguousarray(np.tile(base, (U5 // 65536 + 1, 1))[:U5])
    Q5 = (rng.normal(size=(I5, d)) / d).astype(np.float32)
    Qb5 = np.zeros((I5, 1), np.float32)
    gen_s = time.perf_counter() - t0
    opt = {"evaluation_on_learning": False, "compute_loss_on_training": False, "early_stopping_rounds": 0, "save_best": False,
           "evaluation_period": 5, "save_period": 10, "random_seed": 7, "validation": {}, "accelerator": True, "num_workers": 8,
           "hyper_threads": 256, "num_iters": 10, "d": d, "threshold": 1.0, "score_func": "dot", "max_trials": 500, "update_i": True,
           "update_j": True, "reg_u": 0.0, "reg_i": 0.0, "reg_j": 0.0, "optimizer": "adagrad", "lr": 0.05, "min_lr": 0.0001,
           "beta1": 0.9, "beta2": 0.999, "eps": 1e-10, "per_coordinate_normalize": False, "model_path": "", "data_opt": {}}
    g = CyWARP()
    assert g.init(write_opt(opt))
    g.sync_every_epoch = False
    t0 = time.perf_counter()
    g.initialize_model(P5, Q5, Qb5, keys5.shape[0], True)
    g.set_resident_csr(indptr5, keys5)
    up_s = time.perf_counter() - t0
    ep = []
    for e in range(3):
        g.reset_stats()
        t0 = time.perf_counter()
        g.add_jobs(0, U5, indptr5, None)
        t1 = time.perf_counter()
        g.update_parameters()
        t2 = time.perf_counter()
        st = g.stats()
        ep.append({"epoch": e, "trial_kernel_ms": st["kernel_ms"], "optimizer_ms": st["optimizer_ms"], "wall_ms": (t2 - t0) * 1e3,
                   "positives_per_s": keys5.shape[0] / (t1 - t0), "mean_scored_negatives_T": st["scored_negatives"] / keys5.shape[0],
                   "accepted_frac": st["accepted"] / keys5.shape[0],
                   "optimizer_GBps": (U5 + I5) * d * 4 * 6 / (st["optimizer_ms"] * 1e-3) / 1e9})
        print("warp_c5", ep[-1], flush=True)
    out["warp_10Mx1M_1B_d256_one_gpu"] = {"host_generation_s": gen_s, "upload_s": up_s, "epochs": ep,
                                          "hbm_resident_GB": (3 * (U5 + I5) * d * 4 + keys5.nbytes * 2 + indptr5.nbytes) / 1e9}
    print("warp_c5 resident", out["warp_10Mx1M_1B_d256_one_gpu"]["hbm_resident_GB"], "GB; gen", gen_s, "s; upload", up_s, "s", flush=True)
    del g, P5, Q5, keys5, indptr5

if "topk" in which:
    # the consumer right after training: top-100 of every user over all items (validation / ParALS.topk_recommendation)
    from buffalo_amd import parallel as par
    rng = np.random.default_rng(0)
    d, k = 128, 100
    P = rng.normal(scale=0.1, size=(U, d)).astype(np.float32)
    Q = rng.normal(scale=0.1, size=(I, d)).astype(np.float32)
    eng = par.TopK()
    nob, nop = np.array([[]], np.float32), np.array([], np.int32)
    res = {}
    for nq, fused in ((128, -1), (4096, -1), (U, 0), (U, -1)):
        # fused = 0: dense score buffer + select; -1: the size rule (fused threshold filter from 8192 queries x 8192 items up)
        eng.set_mode("fused", fused)
        idx = np.arange(nq, dtype=np.int32)
        ok, osc = np.empty((nq, k), np.int32), np.empty((nq, k), np.float32)
        eng.dot_topn(idx, P, Q, nob, ok, osc, nop, k)
        eng.reset_stats()
        t0 = time.perf_counter()
        eng.dot_topn(idx, P, Q, nob, ok, osc, nop, k)
        dt = time.perf_counter() - t0
        st = eng.stats()
        name = "nq%d%s" % (nq, "_dense" if (fused == 0) else "")
        res[name] = {"wall_ms_host_arrays": dt * 1e3, "scores_kernel_ms": st["kernel_ms"], "select_and_aux_kernel_ms": st["aux_ms"],
                     "scores_TFLOPs": 2.0 * nq * I * d / (st["kernel_ms"] * 1e-3) / 1e12, "queries_per_s": nq / dt,
                     "rows_redone_densely": st["merges"]}
        if nq == U:
            res[name]["keys_checksum"] = int(ok.astype(np.int64).sum())
        print("topk", name, res[name], flush=True)
    assert res["nq%d" % U]["keys_checksum"] == res["nq%d_dense" % U]["keys_checksum"]
    from oracle import oracle as orc     # CPU restatement of parallel::dot_topn, all host cores (OpenMP), bounded sample
    nq = 4096
    idx = np.arange(nq, dtype=np.int32)
    ok, osc = np.empty((nq, k), np.int32), np.empty((nq, k), np.float32)
    t0 = time.perf_counter()
    orc.dot_topn(idx, P, Q, nob, ok, osc, nop, k)
    dt = time.perf_counter() - t0
    res["cpu_oracle"] = {"queries": nq, "wall_ms": dt * 1e3, "queries_per_s": nq / dt, "cores": os.cpu_count()}
    print("topk cpu", res["cpu_oracle"], flush=True)
    out["topk_ml20m_d128_k100"] = res

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bench_extra.json"), "w"), indent=1)
