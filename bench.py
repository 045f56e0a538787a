#!/usr/bin/env python
"""Headline benchmark: BPRMF training throughput on ML-20M-shaped synthetic interactions, d=128.

    python bench.py --gpus N --steps K --warmup W

A "step" is one epoch of the hot path: one `add_jobs` pass of the fused sampler + BPR update kernel
over every interaction of the (resident) CSR, followed by `update_parameters` (a no-op for sgd).
N=1 runs BASELINE.json configs[1]; N>1 is launched by torch.distributed.run, one rank per GPU, users
sharded, item factors replicated and delta-all-reduced over RCCL once per minibatch (buffalo_amd/dist.py).
Rank 0 prints ONE JSON line (metric/value/roofline/cpu_baseline ...).  At N=1 the line also carries an
`extra` block: the two other inner loops north_star names, measured in the same process after the timed
region -- ALS at BASELINE configs[2] (MFMA utilisation of the Gramian/solve kernel) and WARP at configs[4]'s
d=256 on the ML-20M shape (algorithmic GB/s with the measured T) -- each with its own oracle timing.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3  # same guide: v_mfma_f32_32x32x2_f32, exact fp32 (the type ALS computes in)
MFMA_F16_PEAK_TF = 2500.0     # dense f16/bf16 (MI355X_MICROARCH.md)
D = 128
CPU_KIND = "port"
CPU_WHAT = ("restatement of reference CPU path (oracle/buffalo_oracle.cc, compiled with the reference's flags "
            "-O3 -fopenmp -mavx2 -mfma; the reference's own C++ cannot be built here: Eigen/json11/spdlog submodules are empty)")


def bpr_options(num_iters, seed=7, **kw):
    opt = {  # BPRMFOption defaults (/root/reference/buffalo/algo/options.py:220-252) at d=128
        "evaluation_on_learning": False, "compute_loss_on_training": False, "early_stopping_rounds": 0,
        "save_best": False, "evaluation_period": 100, "save_period": 10, "random_seed": seed,
        "validation": {}, "accelerator": True, "use_bias": True, "num_workers": 8, "hyper_threads": 256,
        "num_iters": num_iters, "d": D, "update_i": True, "update_j": True, "reg_u": 0.025,
        "reg_i": 0.025, "reg_j": 0.025, "reg_b": 0.025, "optimizer": "sgd", "lr": 0.002,
        "min_lr": 0.0001, "beta1": 0.9, "beta2": 0.999, "eps": 1e-10, "per_coordinate_normalize": False,
        "num_negative_samples": 1, "sampling_power": 0.0, "verify_neg": True, "random_positive": False,
        "model_path": "", "data_opt": {},
    }
    opt.update(kw)
    return opt


def write_opt(opt):
    import tempfile
    f = tempfile.NamedTemporaryFile(mode="w", suffix=".json", delete=False)
    json.dump(opt, f)
    f.close()
    return f.name


def load_matrix(shape_name, seed):
    """Synthetic CSR; cached under /tmp so repeated runs on one box skip the ~25 s generation."""
    from buffalo_amd import synth
    U, I, nnz = synth.SHAPES[shape_name]
    cache = "/tmp/bfh_synth_%s_%d.npz" % (shape_name, seed)
    if os.path.exists(cache):
        z = np.load(cache)
        return synth.CSR(U, I, z["indptr"], z["keys"], np.ones(z["keys"].shape[0], np.float32))
    csr = synth.generate(U, I, nnz, seed=seed)
    try:   # written under a private name and renamed: N ranks may get here at the same time
        tmp = "%s.%d.tmp.npz" % (cache, os.getpid())
        np.savez(tmp, indptr=csr.indptr, keys=csr.keys)
        os.replace(tmp, cache)
    except OSError:
        pass
    return csr


def cpu_baseline(csr, target_seconds=12.0):
    """The oracle (restatement of the reference CPU path, reference's compile flags) timed on this
    box's host cores over a bounded prefix of the same workload."""
    from oracle import oracle as orc
    from buffalo_amd import synth
    orc.build()
    cores = os.cpu_count() or 1
    I = csr.num_items

    def run(n_users, workers=cores):
        nnz = int(csr.indptr[n_users - 1])
        opt = bpr_options(1, accelerator=False, num_workers=workers)
        P, Q, Qb = synth.init_factors(n_users, I, D, seed=7)
        o = orc.OracleBPRMF()
        path = write_opt(opt)
        assert o.init(path)
        os.unlink(path)
        o.initialize_model(P, Q, Qb, nnz)
        o.set_cumulative_table(np.zeros(I, np.int64), I)
        o.launch_workers()
        keys = np.ascontiguousarray(csr.keys[:nnz])
        ip = np.ascontiguousarray(csr.indptr[:n_users])
        t0 = time.perf_counter()
        o.add_jobs(0, n_users, ip, keys)
        o.join()                     # returns when every queued job has been processed
        dt = time.perf_counter() - t0
        return nnz, dt

    probe_users = int(np.searchsorted(csr.indptr, 300000)) + 1
    nnz0, dt0 = run(probe_users)
    rate0 = nnz0 / dt0
    want = int(min(csr.nnz, max(nnz0, rate0 * target_seconds)))
    n_users = min(csr.num_users, int(np.searchsorted(csr.indptr, want)) + 1)
    nnz1, dt1 = run(n_users)
    # the reference's own benchmark setting is 8 workers (tests/algo/test_performance.py:53): same port, sized by its own probe
    # (the job queue of the CPU path is contended, so fewer workers can be FASTER than all cores)
    nnz_p, dt_p = run(probe_users, workers=8)
    want8 = int(min(csr.nnz, max(nnz_p, nnz_p / dt_p * 6.0)))
    nnz8, dt8 = run(min(csr.num_users, int(np.searchsorted(csr.indptr, want8)) + 1), workers=8)
    # the headline figure is the 8-worker one -- the reference's own benchmark setting, and the faster of the two (its job queue
    # is contended: all cores are SLOWER); the all-core run is kept beside it
    return {"value": nnz8 / dt8, "unit": "updates/s", "cores": 8, "kind": CPU_KIND, "what": CPU_WHAT,
            "sample": "first %d interactions (1 epoch) of the same matrix, 8 std::thread workers (the reference's benchmark setting, "
                      "tests/algo/test_performance.py:53), %.1f s" % (nnz8, dt8),
            "value_all_cores": nnz1 / dt1, "all_cores": cores,
            "sample_all_cores": "first %d users (%d interactions, 1 epoch), %d workers, %.1f s" % (n_users, nnz1, cores, dt1)}


ALS_OPT = {  # ALSOption defaults (/root/reference/buffalo/algo/options.py:66-86) at d=128 (=> iALS++, Q-13)
    "evaluation_on_learning": False, "compute_loss_on_training": False, "early_stopping_rounds": 0, "save_best": False,
    "evaluation_period": 1, "save_period": 10, "random_seed": 7, "validation": {}, "adaptive_reg": False, "save_factors": False,
    "accelerator": True, "d": D, "num_iters": 10, "num_workers": 8, "hyper_threads": 256, "num_cg_max_iters": 3, "reg_u": 0.1,
    "reg_i": 0.1, "alpha": 8.0, "optimizer": "manual_cg", "cg_tolerance": 1e-10, "block_size": 32, "eps": 1e-10,
    "model_path": "", "data_opt": {}}
WARP_D = 256
WARP_OPT = {  # WARPOption defaults (options.py:286-311) at configs[4]'s d=256
    "evaluation_on_learning": False, "compute_loss_on_training": False, "early_stopping_rounds": 0, "save_best": False,
    "evaluation_period": 5, "save_period": 10, "random_seed": 7, "validation": {}, "accelerator": True, "num_workers": 8,
    "hyper_threads": 256, "num_iters": 10, "d": WARP_D, "threshold": 1.0, "score_func": "dot", "max_trials": 500, "update_i": True,
    "update_j": True, "reg_u": 0.0, "reg_i": 0.0, "reg_j": 0.0, "optimizer": "adagrad", "lr": 0.05, "min_lr": 0.0001,
    "beta1": 0.9, "beta2": 0.999, "eps": 1e-10, "per_coordinate_normalize": False, "model_path": "", "data_opt": {}}


def _opt_file(opt):
    path = write_opt(opt)
    return path


def extra_als(csr, seed, epochs=5, cpu=True):
    """BASELINE configs[2]: ALS (iALS++ at d=128) on the ML-20M shape, one GPU; both CSR orientations resident."""
    from buffalo_amd import ingest, synth
    from buffalo_amd.backend import CyALS
    U, I, nnz = csr.num_users, csr.num_items, csr.nnz
    rng = np.random.default_rng(seed)
    vals = (1 + rng.poisson(1.0, size=nnz)).astype(np.float32)      # counts 1 + Poisson(1)
    col = ingest.coo_to_csr(csr.keys, csr.rows(), vals, I, U)       # colwise orientation, built on the device (bfh_coo_to_csr)
    P, Q, _ = synth.init_factors(U, I, D, seed=seed)
    g = CyALS()
    path = _opt_file(ALS_OPT)
    assert g.init(path)
    os.unlink(path)
    g.initialize_model(P, Q)
    g.set_resident_csr(0, csr.indptr, csr.keys, vals)
    g.set_resident_csr(1, col["indptr"], col["key"], col["val"])
    g.set_mode("als_writeback", 0)

    def epoch():
        g.precompute(0)
        g.partial_update(0, U, csr.indptr, None, None, 0)
        g.precompute(1)
        g.partial_update(0, I, col["indptr"], None, None, 1)
    epoch()
    g.reset_stats()
    t0 = time.perf_counter()
    for _ in range(epochs):
        epoch()
    dt = (time.perf_counter() - t0) / epochs
    st = g.stats()
    T = D // 32
    gram_flop = 2 * nnz * (T * (T + 1) // 2) * 2 * 32 * 32          # the upper-triangle tiles of both half-epochs, once (what the fp32 instruction issued)
    mfma_flop = 3 * gram_flop                                       # issued now: x = h + l in f16, the three products hh + hl + lh (als_gram_kernel<SPLIT>)
    kernel_s = st["kernel_ms"] / epochs * 1e-3
    alg_bytes = 2 * nnz * (4 * D + 8) + (U + I) * (8 * D + 8) + (U + I) * 4 * D     # SURVEY 8(d) B_als, both half-epochs
    out = {"config": "ALS iALS++ (block 32, 3 CG steps), ml20m-shaped synthetic (%d x %d, %d nnz, values 1+Poisson(1)), d=%d, f32, "
                     "rowwise + colwise CSR and factors resident in HBM" % (U, I, nnz, D),
           "epoch_ms": dt * 1e3, "interactions_per_s": 2 * nnz / dt, "kernel": "als_pc_kernel (producer / consumer wave pairs: gather + residuals + f16 cut | Gramian on the f16 matrix cores at fp32 accuracy + in-register block CG)",
           "kernel_ms_per_epoch": kernel_s * 1e3, "gramian_ff_ms_per_epoch": st["aux_ms"] / epochs,
           "mfma": {"issued_TFLOPs": mfma_flop / kernel_s / 1e12, "peak_TFLOPs": MFMA_F16_PEAK_TF,
                    "frac": mfma_flop / kernel_s / 1e12 / MFMA_F16_PEAK_TF,
                    "instruction": "v_mfma_f32_32x32x16_f16, three per tile and 16 entries (split-f16 pass; fp32 accuracy)",
                    "issued_flop_per_epoch": mfma_flop,
                    "gramian_TFLOPs": gram_flop / kernel_s / 1e12,
                    # what a block-diagonal formulation would need, 4 nnz d bs + 2 (U + I) d^2 -- the "useful" flops
                    "useful_flop_per_epoch": 4 * nnz * D * 32 + 2 * (U + I) * D * D,
                    "useful_frac_of_fp32_peak": (4 * nnz * D * 32 + 2 * (U + I) * D * D) / kernel_s / 1e12 / MFMA_F32_PEAK_TF,
                    "note": "not bound by the matrix cores: the gather skeleton alone is 2.5 ms per epoch (~8 TB/s out of L2 / Infinity Cache), the "
                            "producers' arithmetic and the user half's per-row VALU work (block solve, handshakes) make up the rest "
                            "(DESIGN 4.5, profiles/r04_micro_simd_overlap.txt); the same Gramian through v_mfma_f32_32x32x2_f32 (als_split_f16=0) needs %.1f ms of matrix-core "
                            "time alone at its %.0f TFLOP/s peak" % (gram_flop / MFMA_F32_PEAK_TF / 1e9, MFMA_F32_PEAK_TF)},
           "hbm": {"algorithmic_bytes_per_epoch": alg_bytes, "achieved_GBps": alg_bytes / kernel_s / 1e9,
                   "frac": alg_bytes / kernel_s / 1e9 / HBM_PEAK_GBS}}
    del g
    if cpu:
        from oracle import oracle as orc
        orc.build()
        cores = os.cpu_count() or 1

        def run(n_users):
            Po, Qo, _ = synth.init_factors(n_users, I, D, seed=seed)
            o = orc.OracleALS()
            path = _opt_file(dict(ALS_OPT, accelerator=False, num_workers=cores))
            assert o.init(path)
            os.unlink(path)
            o.initialize_model(Po, Qo)
            m = int(csr.indptr[n_users - 1])
            ip, k, v = np.ascontiguousarray(csr.indptr[:n_users]), np.ascontiguousarray(csr.keys[:m]), np.ascontiguousarray(vals[:m])
            t0 = time.perf_counter()
            o.precompute(0)
            o.partial_update(0, n_users, ip, k, v, 0)
            return m, time.perf_counter() - t0
        m0, d0 = run(int(np.searchsorted(csr.indptr, 200000)) + 1)
        want = int(min(nnz, max(m0, m0 / d0 * 6.0)))
        m1, d1 = run(min(U, int(np.searchsorted(csr.indptr, want)) + 1))
        out["cpu_baseline"] = {"value": m1 / d1, "unit": "interactions/s", "cores": cores, "kind": CPU_KIND, "what": CPU_WHAT,
                               "sample": "user half-epoch (precompute + partial_update, iALS++) over the first %d interactions of the same "
                                         "matrix, OpenMP %d threads, %.1f s" % (m1, cores, d1)}
    return out


def extra_als_wide(csr, seed, d, epochs=3):
    """ALS at 128 < d <= 256 (iALS++, block 32) on the ML-20M shape: als_wide_kernel -- the row's tiles spread over ceil(T/2) waves,
    fp32 matrix instruction, residual-first gradient."""
    from buffalo_amd import ingest, synth
    from buffalo_amd.backend import CyALS
    U, I, nnz = csr.num_users, csr.num_items, csr.nnz
    vals = (1 + np.random.default_rng(seed).poisson(1.0, size=nnz)).astype(np.float32)
    col = ingest.coo_to_csr(csr.keys, csr.rows(), vals, I, U)
    P, Q, _ = synth.init_factors(U, I, d, seed=seed)
    g = CyALS()
    path = _opt_file(dict(ALS_OPT, d=d))
    assert g.init(path)
    os.unlink(path)
    g.initialize_model(P, Q)
    g.set_resident_csr(0, csr.indptr, csr.keys, vals)
    g.set_resident_csr(1, col["indptr"], col["key"], col["val"])
    g.set_mode("als_writeback", 0)

    def epoch():
        g.precompute(0)
        g.partial_update(0, U, csr.indptr, None, None, 0)
        g.precompute(1)
        g.partial_update(0, I, col["indptr"], None, None, 1)
    epoch()
    g.reset_stats()
    t0 = time.perf_counter()
    for _ in range(epochs):
        epoch()
    dt = (time.perf_counter() - t0) / epochs
    st = g.stats()
    T = d // 32
    gram_flop = 2 * nnz * (T * (T + 1) // 2) * 2 * 32 * 32
    kernel_s = st["kernel_ms"] / epochs * 1e-3
    return {"config": "ALS iALS++ (block 32), ml20m-shaped synthetic, d=%d, f32: als_wide_kernel" % d, "epoch_ms": dt * 1e3,
            "kernel_ms_per_epoch": kernel_s * 1e3, "interactions_per_s": 2 * nnz / dt,
            "mfma_fp32_issued_TFLOPs": gram_flop / kernel_s / 1e12, "mfma_fp32_frac": gram_flop / kernel_s / 1e12 / MFMA_F32_PEAK_TF}


def warp_epoch_row(st, nnz, d, U, I, wall_s, presample=4, chunk_runs=None):
    """One WARP epoch's numbers from the backend's counters.  Three byte figures, never mixed:
    * algorithmic (the reference formulation warp.cc:135-165): per accepted positive (8 + T) rows of 4d bytes + key,
      per rejected one (2 + T) rows + key -- 6 of the 8 are the read-modify-writes of three gradient rows;
    * implemented model (DESIGN "WARP"): what the kernels of csrc/warp.hip + the sorted gather move by construction -- the trial
      kernel reads Q[pos] and every candidate row it fetched (`loaded_rows`: scored + speculated) once, P[u] / gradP once per
      user run; the item-side gradient rows are NOT read-modify-written per positive but summed by grad_gather_kernel over
      item-sorted incidence lists (one P[u] row read per accepted incidence and list, one gradQ row written per item and list);
    * counter traffic comes from the rocprofv3 --pmc passes (profiles/, scripts/pmc_kernels.py), not from here."""
    acc, scored, loaded = st["accepted"], st["scored_negatives"], st.get("loaded_rows", 0) or st["scored_negatives"]
    row = 4 * d
    alg = (8 * acc + 2 * (nnz - acc) + scored) * row + 4 * nnz
    S = presample
    runs = chunk_runs if chunk_runs is not None else U + nnz // 256
    impl = {
        "presample": nnz * (4 + 4 * S + 4),
        "trial_kernel": nnz * (8 + 4 * S + 4 + 8) + (nnz + loaded) * row + runs * 3 * row,
        "sort": nnz * 16 * 3,                                  # (key, index) pairs, ~3 radix passes, read + write
        "gather": 2 * (acc * (row + 16) + I * 3 * row),        # two lists: P[u] per incidence; Q row + gradQ RMW per item
        "optimizer": (U + I) * 6 * row,
    }
    dev_s = (st["kernel_ms"] + st["aux_ms"]) * 1e-3
    dev_all_s = dev_s + st["optimizer_ms"] * 1e-3
    return {"epoch_ms": wall_s * 1e3, "trial_kernel_ms": st["kernel_ms"], "sort_and_gather_ms": st["aux_ms"], "optimizer_ms": st["optimizer_ms"],
            "positives_per_s": nnz / wall_s, "mean_scored_negatives_T": scored / nnz, "candidate_rows_fetched_per_positive": loaded / nnz,
            "accepted_frac": acc / nnz, "algorithmic_bytes": alg, "algorithmic_GBps": alg / dev_s / 1e9, "hbm_frac": alg / dev_s / 1e9 / HBM_PEAK_GBS,
            "implemented_model_bytes": impl, "implemented_model_total": sum(impl.values()),
            "implemented_model_GBps": sum(impl.values()) / dev_all_s / 1e9,
            "implemented_model_frac": sum(impl.values()) / dev_all_s / 1e9 / HBM_PEAK_GBS,
            "optimizer_GBps": impl["optimizer"] / max(st["optimizer_ms"] * 1e-3, 1e-12) / 1e9}


def _warp_counter_traffic(key):
    """Counter traffic per epoch of the WARP kernels from the committed rocprofv3 --pmc passes (same command, scripts/gpu_profile_warp.sh)."""
    try:
        with open(os.path.join(ROOT, "profiles", "warp_pmc_latest.json")) as f:
            return json.load(f).get(key)
    except (OSError, ValueError):
        return None


def _warp_cpu_baseline(indptr, keys, I, d, seed, start_entries=200000, seconds=6.0):
    from buffalo_amd import synth
    from oracle import oracle as orc
    orc.build()
    cores = os.cpu_count() or 1
    nnz = int(keys.shape[0])
    U = int(indptr.shape[0])

    def run(n_users):
        m = int(indptr[n_users - 1])
        Po, Qo, Qbo = synth.init_factors(n_users, I, d, seed=seed, signed=True)
        o = orc.OracleWARP()
        path = _opt_file(dict(WARP_OPT, accelerator=False, num_workers=cores, num_iters=1))
        assert o.init(path)
        os.unlink(path)
        o.initialize_model(Po, Qo, Qbo, m)
        o.set_cumulative_table(np.zeros(I, np.int64), I)
        o.launch_workers()
        ip, k = np.ascontiguousarray(indptr[:n_users]), np.ascontiguousarray(keys[:m])
        t0 = time.perf_counter()
        o.add_jobs(0, n_users, ip, k)
        o.join()
        return m, time.perf_counter() - t0
    m0, d0 = run(int(np.searchsorted(indptr, start_entries)) + 1)
    want = int(min(nnz, max(m0, m0 / d0 * seconds)))
    m1, d1 = run(min(U, int(np.searchsorted(indptr, want)) + 1))
    return {"value": m1 / d1, "unit": "positives/s", "cores": cores, "kind": CPU_KIND, "what": CPU_WHAT,
            "sample": "first %d interactions of the same matrix (%.3g of it), 1 epoch (add_jobs .. join), %d std::thread workers, %.1f s"
                      % (m1, m1 / nnz, cores, d1)}


def _warp_epochs(g, U, indptr, nnz, d, I, epochs, until_T=None, max_epochs=None):
    """`epochs` epochs; with `until_T`, further ones (at most `max_epochs` in all) until the trial loop scores that many negatives
    per positive -- the regime training lives in, not the first epochs' T = 1 where every first draw violates the margin."""
    eps = []
    e = 0
    while e < epochs or (until_T is not None and e < (max_epochs or epochs) and eps[-1]["mean_scored_negatives_T"] < until_T):
        g.reset_stats()
        t0 = time.perf_counter()
        g.add_jobs(0, U, indptr, None)
        g.update_parameters()
        dt = time.perf_counter() - t0
        eps.append(warp_epoch_row(g.stats(), nnz, d, U, I, dt))
        e += 1
    return eps


def _warp_summary(eps, out):
    """The epoch to quote is the LAST one: epochs 1-2 start from near-zero factors where every first negative violates the margin
    (T = 1, everything accepted -- the easy regime); by the third the trial loop rejects (T ~ 4 on the ML-20M shape)."""
    last = eps[-1]
    out.update({"epochs": eps, "quoted_epoch": len(eps) - 1, "epoch_ms": last["epoch_ms"], "positives_per_s": last["positives_per_s"],
                "mean_scored_negatives_T": last["mean_scored_negatives_T"], "algorithmic_GBps": last["algorithmic_GBps"], "hbm_frac": last["hbm_frac"],
                "implemented_model_frac": last["implemented_model_frac"]})
    return out


def extra_warp(csr, seed, epochs=3, cpu=True):
    """WARP (warp.cc:103-201) at BASELINE configs[4]'s d=256 / adagrad on the ML-20M shape, one GPU."""
    from buffalo_amd import synth
    from buffalo_amd.backend import CyWARP
    U, I, nnz = csr.num_users, csr.num_items, csr.nnz
    d = WARP_D
    P, Q, Qb = synth.init_factors(U, I, d, seed=seed, signed=True)
    Qb *= 0
    g = CyWARP()
    path = _opt_file(WARP_OPT)
    assert g.init(path)
    os.unlink(path)
    g.sync_every_epoch = False
    g.initialize_model(P, Q, Qb, nnz, True)
    g.set_resident_csr(csr.indptr, csr.keys)
    g.add_jobs(0, U, csr.indptr, None)      # warm-up epoch (allocations, the positive incidence list)
    g.update_parameters()
    eps = _warp_epochs(g, U, csr.indptr, nnz, d, I, epochs)
    out = _warp_summary(eps, {
        "config": "WARP adagrad, dot score, max_trials 500, ml20m-shaped synthetic (%d x %d, %d nnz), d=%d, f32, CSR + factors + "
                  "optimizer state resident in HBM" % (U, I, nnz, d),
        "kernels": "warp_presample_kernel + warp_update_kernel (trial loop, gradP in registers) + radix sort of the accepted negatives + "
                   "grad_gather_kernel x2 + sgd_update_rows_kernel (adagrad + unit-ball projection)"})
    tr = _warp_counter_traffic("ml20m")
    if tr:
        out["counter_traffic"] = tr
    del g
    if cpu:
        out["cpu_baseline"] = _warp_cpu_baseline(csr.indptr, csr.keys, I, d, seed)
    return out


def warp_c5_inputs(skew=True):
    """BASELINE configs[4]'s shape for ONE GPU: 10 M users x 1 M items, 1 B interactions, d=256.  Every user has 100 items, one in
    each 10,000-wide band of the catalogue (sorted keys, no duplicates).  `skew`: the item inside a band is floor(10^4 x^3) for a
    hashed x in [0, 1) -- a popularity law (the head item of a band is chosen by 4.6 % of the users), so that there is something to
    rank and the trial loop leaves the T = 1 regime as it does on real data; without it (round 2's generator) every epoch accepts
    the first draw.  Factors signed N(0, 1/d^2) (Q-18), P tiled from 65,536 distinct rows to keep host generation to seconds.
    41.9 GB resident (P, Q, gradients, adagrad state, keys, row ids)."""
    U5, I5, deg, d = 10_000_000, 1_000_000, 100, WARP_D
    step = I5 // deg
    keys = np.empty((U5, deg), dtype=np.int32)
    band_hash = ((np.arange(deg, dtype=np.int64) * 104729) % step).astype(np.int32)
    band_base = np.arange(deg, dtype=np.int32) * step
    for u0 in range(0, U5, 500_000):              # int32 throughout: ~11 s for the 10^9 keys on the host
        u = np.arange(u0, min(U5, u0 + 500_000), dtype=np.int64)
        hu = ((u * 7919) % step).astype(np.int32)
        if skew:
            h = hu[:, None] + band_hash[None, :]
            np.subtract(h, step, out=h, where=h >= step)
            off = h * h
            off //= step
            off *= h
            off //= step                          # ~ floor(step x^3), x = h / step
        else:
            off = np.repeat(hu[:, None], deg, axis=1)
        off += band_base[None, :]
        keys[u0:u0 + u.shape[0]] = off
    keys = keys.reshape(-1)
    indptr = (np.arange(U5, dtype=np.int64) + 1) * deg
    rng = np.random.default_rng(7)
    base = (rng.normal(size=(65536, d)) / d).astype(np.float32)
    P = np.ascontiguousarray(np.tile(base, (U5 // 65536 + 1, 1))[:U5])
    Q = (rng.normal(size=(I5, d)) / d).astype(np.float32)
    Qb = np.zeros((I5, 1), np.float32)
    return indptr, keys, P, Q, Qb


def extra_warp_c5(seed, epochs=6, cpu=True):
    """BASELINE configs[4] (10 M x 1 M, 1 B nnz, d=256 -- the config shards the users over 8 GPUs) on ONE GPU: everything fits
    the 288 GB of one MI355X.  CPU baseline: the oracle on the first 1/100 of the interactions."""
    from buffalo_amd.backend import CyWARP
    t0 = time.perf_counter()
    indptr, keys, P, Q, Qb = warp_c5_inputs()
    gen_s = time.perf_counter() - t0
    U, I, nnz, d = P.shape[0], Q.shape[0], int(keys.shape[0]), WARP_D
    g = CyWARP()
    path = _opt_file(WARP_OPT)
    assert g.init(path)
    os.unlink(path)
    g.sync_every_epoch = False
    t0 = time.perf_counter()
    g.initialize_model(P, Q, Qb, nnz, True)
    g.set_resident_csr(indptr, keys)
    up_s = time.perf_counter() - t0
    eps = _warp_epochs(g, U, indptr, nnz, d, I, epochs, until_T=3.0, max_epochs=12)
    out = _warp_summary(eps, {
        "config": "WARP adagrad, dot score, max_trials 500, configs[4]-shaped synthetic (%d x %d, %d nnz), d=%d, f32, ONE GPU, everything "
                  "resident in HBM" % (U, I, nnz, d),
        "host_generation_s": gen_s, "upload_s": up_s,
        "hbm_resident_GB": (3 * (U + I) * d * 4 + keys.nbytes * 2 + indptr.nbytes) / 1e9})
    tr = _warp_counter_traffic("c5")
    if tr:
        out["counter_traffic"] = tr
    del g
    if cpu:
        n100 = U // 100
        out["cpu_baseline"] = _warp_cpu_baseline(indptr[:n100], keys[:int(indptr[n100 - 1])], I, d, seed, seconds=4.0)
        out["cpu_baseline"]["sample"] += "; the first 1/100 of the users against the full item table, to be scaled linearly (SURVEY 8(d))"
    return out


def extra_topk(csr, seed, cpu=True):
    """The consumer right after training (parallel::dot_topn, _core.hpp:89-142): top-100 of every user over all items, host arrays
    in and out -- the fused path of DESIGN 4.6 (thresholds from a column sample, filtered MFMA sweep, wave-per-row selection)."""
    from buffalo_amd import parallel as par
    U, I = csr.num_users, csr.num_items
    rng = np.random.default_rng(seed)
    k = 100
    P = rng.normal(scale=0.1, size=(U, D)).astype(np.float32)
    Q = rng.normal(scale=0.1, size=(I, D)).astype(np.float32)
    eng = par.TopK()
    nob, nop = np.array([[]], np.float32), np.array([], np.int32)
    idx = np.arange(U, dtype=np.int32)
    ok, osc = np.empty((U, k), np.int32), np.empty((U, k), np.float32)
    out = {"config": "dot_topn, %d queries x %d candidates, d=%d, k=%d, f32, host arrays in -> host arrays out" % (U, I, D, k)}
    for name, fused in (("dense_score_buffer", 0), ("fused", -1)):
        eng.set_mode("fused", fused)
        eng.dot_topn(idx, P, Q, nob, ok, osc, nop, k)
        eng.reset_stats()
        t0 = time.perf_counter()
        eng.dot_topn(idx, P, Q, nob, ok, osc, nop, k)
        dt = time.perf_counter() - t0
        st = eng.stats()
        out[name] = {"wall_ms": dt * 1e3, "queries_per_s": U / dt, "score_kernels_ms": st["kernel_ms"], "select_and_aux_kernels_ms": st["aux_ms"],
                     "mfma_TFLOPs": 2.0 * U * I * D / (st["kernel_ms"] * 1e-3) / 1e12, "mfma_frac": 2.0 * U * I * D / (st["kernel_ms"] * 1e-3) / 1e12 / MFMA_F32_PEAK_TF,
                     "rows_redone_densely": st["merges"], "keys_checksum": int(ok.astype(np.int64).sum())}
    out["identical"] = out["fused"]["keys_checksum"] == out["dense_score_buffer"]["keys_checksum"]
    del eng
    if cpu:
        from oracle import oracle as orc
        orc.build()
        nq = 4096
        ok2, os2 = np.empty((nq, k), np.int32), np.empty((nq, k), np.float32)
        t0 = time.perf_counter()
        orc.dot_topn(np.ascontiguousarray(idx[:nq]), P, Q, nob, ok2, os2, nop, k)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": nq / dt, "unit": "queries/s", "cores": os.cpu_count() or 1, "kind": CPU_KIND, "what": CPU_WHAT,
                               "sample": "the first %d queries of the same sweep, OpenMP, %.2f s" % (nq, dt),
                               "agrees_with_device": bool((ok2 == ok[:nq]).mean() > 0.999)}
    return out


def extra_sppmi(csr, seed, cpu=True):
    """CoFactor's context input: the SPPMI matrix of the matrix read as a stream (stream.py:257-267 + fileio.hpp:109-254 + stream.py:169-195),
    windows 5, k 1 -- built in HBM by bfh_sppmi_* (DESIGN 4.7b)."""
    from buffalo_amd import ingest
    rng = np.random.default_rng(seed)
    rows = csr.rows()
    order = np.argsort(rows + rng.random(csr.nnz))            # a sequence order per user (the matrix stores the items sorted)
    items = np.ascontiguousarray(csr.keys[order])
    g, st = ingest.build_sppmi(csr.indptr, items, csr.num_items, 5, 1, with_stats=True)
    t0 = time.perf_counter()
    g, st = ingest.build_sppmi(csr.indptr, items, csr.num_items, 5, 1, with_stats=True)
    dt = time.perf_counter() - t0
    lines = g["total_lines"]
    out = {"config": "SPPMI of %d sequences / %d events over %d items, windows 5, shift k 1; host stream in -> host (indptr, key, val) out"
                     % (csr.num_users, csr.nnz, csr.num_items),
           "pair_lines": lines, "distinct_pairs": st["launches"], "nnz": int(len(g["key"])), "device_ms": st["kernel_ms"], "wall_ms": dt * 1e3,
           "lines_per_s_device": lines / (st["kernel_ms"] * 1e-3),
           }
    kb = 4 if csr.num_items ** 2 <= 2 ** 32 else 8
    passes = -(-max(1, int(np.ceil(np.log2(float(csr.num_items) ** 2)))) // 8)
    alg = lines * kb * (2 + 2 * passes)
    out["hbm"] = {"algorithmic_bytes": alg, "note": "%d-byte keys: written once, %d radix passes (read + write), read once by the run-length encode"
                                                    % (kb, passes),
                  "achieved_GBps": alg / (st["kernel_ms"] * 1e-3) / 1e9, "frac": alg / (st["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if cpu:
        from oracle import oracle as orc
        orc.build()
        n_users = int(np.searchsorted(csr.indptr, 1000000)) + 1
        ip = np.ascontiguousarray(csr.indptr[:n_users])
        it = np.ascontiguousarray(items[:int(ip[-1])])
        t0 = time.perf_counter()
        o = orc.build_sppmi(ip, it, csr.num_items, 5, 1)
        dt = time.perf_counter() - t0
        port = {"value": o["total_lines"] / dt, "unit": "pair lines/s", "cores": 1, "kind": CPU_KIND, "what": CPU_WHAT +
                " -- in memory: the reference additionally writes, sorts and re-reads three text files",
                "sample": "the first %d sequences (%d events, %d lines), one thread, %.1f s" % (n_users, int(ip[-1]), o["total_lines"], dt)}
        out["cpu_baseline"] = port
        try:   # the reference's OWN compiled builder where oracle/_ref travelled with the snapshot (built from /root/reference by build())
            from oracle import ref_fileio as rf
            if os.path.exists(rf._LIB_PATH):
                workers = os.cpu_count() or 1
                r = rf.timed_build_sppmi(ip, it, csr.num_items, 5, 1, workers)
                if r["nnz"] != len(o["key"]):
                    raise RuntimeError("reference builder: %d entries, oracle: %d" % (r["nnz"], len(o["key"])))
                out["cpu_baseline"] = {"value": r["total_lines"] / r["total_s"], "unit": "pair lines/s", "cores": workers, "kind": "reference",
                                       "what": "the reference's own buffalo/data/fileio.hpp compiled from its source (oracle/_ref) + sort(1), as "
                                               "StreamData._build_sppmi runs them: sort the pair lines, _parallel_build_sppmi, sort its output, "
                                               "_chunking_into_bins; the Python loop that writes the pair lines (stream.py:257-267) is not timed",
                                       "sample": "the first %d sequences (%d events, %d lines), %d workers, %.1f s" % (n_users, int(ip[-1]), r["total_lines"], workers, r["total_s"]),
                                       "stages_s": {k: r[k] for k in ("sort_lines_s", "build_s", "sort_output_s", "chunk_s")}}
                out["cpu_baseline_port"] = port
        except Exception as e:
            out["cpu_baseline_reference_error"] = "%s: %s" % (type(e).__name__, e)
    return out


def extra_ingest(csr, seed, cpu=True):
    """COO records -> compressed rows on the device (bfh_coo_to_csr) -- the matrix's records in shuffled order,
    host arrays in, host (indptr, key, val) out -- beside the reference's own compiled sorter on a sample."""
    from buffalo_amd import ingest
    rng = np.random.default_rng(seed)
    perm = rng.permutation(csr.nnz)
    rows = np.ascontiguousarray(csr.rows()[perm].astype(np.int32))
    cols = np.ascontiguousarray(csr.keys[perm])
    vals = np.ascontiguousarray(csr.vals[perm])
    ingest.coo_to_csr(rows, cols, vals, csr.num_users, csr.num_items, with_stats=True)
    t0 = time.perf_counter()
    g, st = ingest.coo_to_csr(rows, cols, vals, csr.num_users, csr.num_items, with_stats=True)
    dt = time.perf_counter() - t0
    assert np.array_equal(g["indptr"], csr.indptr) and np.array_equal(g["key"], csr.keys)       # the shuffle is undone
    out = {"config": "%d shuffled (row, col, val) records -> rowwise group of %d x %d; host arrays in -> host arrays out" % (csr.nnz, csr.num_users, csr.num_items),
           "records": csr.nnz, "device_ms": st["kernel_ms"], "wall_ms": dt * 1e3, "records_per_s_device": csr.nnz / (st["kernel_ms"] * 1e-3)}
    if cpu:
        try:
            from oracle import ref_fileio as rf
            if os.path.exists(rf._LIB_PATH):
                n = min(csr.nnz, 3000000)
                workers = os.cpu_count() or 1
                r = rf.timed_sort_and_compressed_binarization(rows[:n], cols[:n], vals[:n], csr.num_users, 1, workers)
                assert np.array_equal(r["indptr"], np.cumsum(np.bincount(rows[:n], minlength=csr.num_users)))
                out["cpu_baseline"] = {"value": n / r["total_s"], "unit": "records/s", "cores": workers, "kind": "reference",
                                       "what": "the reference's own buffalo/data/fileio.hpp compiled from its source (oracle/_ref): "
                                               "_sort_and_compressed_binarization, i.e. parse the working text file, stable parallel sort by "
                                               "(row, col), indptr, binary chunks",
                                       "sample": "the first %d shuffled records, %d workers, %.1f s" % (n, workers, r["total_s"])}
        except Exception as e:
            out["cpu_baseline_reference_error"] = "%s: %s" % (type(e).__name__, e)
    return out


def measured_stream_bandwidth(n_bytes=1 << 30, reps=20):
    """What this box's HBM delivers to a trivial kernel ("always also report the fraction of measured
    triad bandwidth"): STREAM triad a = b + s * c (two reads + one write per element) and a plain copy over 1 GiB fp32 arrays,
    one fused elementwise launch each, timed with events on the stream they run on."""
    import torch
    n = n_bytes // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda")
    b = torch.ones_like(a)
    c = torch.ones_like(a)

    def rate(fn, moved):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return moved * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9

    out = {"triad_GBps": rate(lambda: torch.add(b, c, alpha=0.5, out=a), 3 * n_bytes), "copy_GBps": rate(lambda: a.copy_(b), 2 * n_bytes),
           "bytes_per_array": n_bytes, "what": "a = b + 0.5 c and a = b over 1 GiB fp32 arrays, %d launches each" % reps}
    del a, b, c
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=250)   # ~2.2 s timed region on one MI355X
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--shape", default="ml20m")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--minibatches", type=int, default=1, help="all-reduce points per epoch (N>1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the ALS / WARP secondary measurements (N=1)")
    ap.add_argument("--mode", action="append", default=[], help="backend knob name=value (e.g. hogwild_atomic=0)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from buffalo_amd import synth
    from buffalo_amd.backend import CyBPR
    from buffalo_amd.dist import shard_csr

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # test hooks (1-GPU boxes): BFH_DEVICE_OVERRIDE pins every rank to one device, BFH_DIST_BACKEND=gloo
    # replaces RCCL (which refuses two ranks on one GPU); the driver's runs set neither
    if "BFH_DEVICE_OVERRIDE" in os.environ:
        local_rank = int(os.environ["BFH_DEVICE_OVERRIDE"])
    torch.cuda.set_device(local_rank)
    comm_mode = "none"
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("BFH_DIST_BACKEND", "nccl")
        # torch.distributed is the control plane only (rendezvous, barrier, max of the elapsed times: gloo on CPU tensors);
        # the data plane is the library's own communicator (bfh_comm_*: RCCL over xGMI).  Test hooks for one-GPU boxes:
        # BFH_DIST_BACKEND=gloo + BFH_COMM_TRANSPORT=shm + BFH_DEVICE_OVERRIDE=0 run N ranks on one device through the
        # library's shared-memory test transport (RCCL refuses two ranks on one GPU).  BFH_COMM=torch selects the emergency
        # path below that all-reduces the backend's buffers through torch.distributed (bench-only; not the product).
        if backend == "nccl":
            dist.init_process_group("cpu:gloo,cuda:nccl")
        else:
            dist.init_process_group(backend)
        comm_mode = os.environ.get("BFH_COMM", "library")
    assert world == args.gpus, "WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus)

    csr = load_matrix(args.shape, args.seed)
    U, I, nnz = csr.num_users, csr.num_items, csr.nnz
    if world > 1 and args.scaling == "strong":
        u0, u1, ip, keys, nnz_off = shard_csr(csr.indptr, csr.keys, rank, world)
        total_nnz = nnz
    else:
        # weak scaling: every rank trains its own ML-20M-shaped user population against shared items
        u0, u1, ip, keys, nnz_off = 0, U, csr.indptr, csr.keys, rank * nnz
        total_nnz = nnz * world
    n_local_users = u1 - u0
    local_nnz = int(keys.shape[0])

    steps, warmup = args.steps, args.warmup
    opt = bpr_options(num_iters=steps + warmup, seed=args.seed)
    P, Q, Qb = synth.init_factors(U, I, D, seed=args.seed)
    P = np.ascontiguousarray(P[u0:u1])

    obj = CyBPR()
    obj.set_device(local_rank)
    path = write_opt(opt)
    assert obj.init(path)
    os.unlink(path)
    obj.sync_every_epoch = False           # keep the model in HBM inside the timed region
    hog = "3"                               # the backend's default for sgd (bfh_bpr_set_mode "hogwild_atomic")
    knobs = {}
    for kv in args.mode:
        k, v = kv.split("=")
        knobs[k] = v
        obj.set_mode(k, int(v))
        if k == "hogwild_atomic":
            hog = v
    obj.initialize_model(P, Q, Qb, total_nnz, True)
    obj.set_cumulative_table(np.zeros(I, np.int64), I)
    obj.set_resident_csr(ip, keys)        # inputs resident in HBM before the timed region starts
    obj.set_shard(nnz_off, world)
    comm, comm_note = None, None
    if comm_mode == "library":
        from buffalo_amd.backend import Comm
        ok = 1
        try:
            uid = torch.frombuffer(bytearray(Comm.unique_id() if rank == 0 else bytes(128)), dtype=torch.uint8).clone()
            dist.broadcast(uid, src=0)                       # CPU tensor -> gloo
            comm = Comm(world, rank, bytes(uid.numpy().tobytes()), local_rank)
            comm.self_test()
        except Exception as e:                                # every rank must take the same path
            ok, comm_note = 0, "%s: %s" % (type(e).__name__, e)
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            obj.set_comm(comm)
        else:
            comm, comm_mode = None, "torch"
            comm_note = "library communicator unavailable (%s): fell back to torch.distributed" % comm_note
    fallback = None
    if world > 1 and comm_mode == "torch":
        # emergency path (the library's communicator could not be built on this node): the blocking delta all-reduce
        # T <- Z + sum_r (T_r - Z) on the backend's Q / Qb through torch.distributed.  Not the product path; labelled in `config`.
        tq, tb = obj.device_tensor("Q", (I, D)), obj.device_tensor("Qb", (I,))
        fallback = [(t, torch.empty_like(t)) for t in (tq, tb)]
    edges = np.linspace(0, n_local_users, (args.minibatches if world > 1 else 1) + 1).astype(int)

    def step():
        for a, b in zip(edges[:-1], edges[1:]):
            if fallback is not None:
                for t, z in fallback:
                    z.copy_(t)
                torch.cuda.current_stream().synchronize()
            obj.add_jobs(int(a), int(b), ip, None)          # an empty chunk still enters the call's collectives
            if fallback is not None:
                torch.cuda.synchronize()
                for t, z in fallback:
                    t.sub_(z)
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
                    t.add_(z)
                torch.cuda.current_stream().synchronize()
        obj.update_parameters()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.all_reduce(torch.zeros(1))                  # CPU tensor: a gloo barrier that never touches the data plane
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    obj.reset_stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if comm is not None:
        obj.comm_flush()                                     # the exchange still in flight belongs to the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = obj.stats()
    breakdown = None
    if world > 1:
        # per step, the slowest rank of each: where a scaling run's time goes (device times from HIP events inside the library)
        b = torch.tensor([st["kernel_ms"], st["aux_ms"], st["exchange_kernel_ms"], st["allreduce_ms"], float(local_nnz)], dtype=torch.float64)
        lo = b.clone()
        dist.all_reduce(b, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        k = 1.0 / max(steps, 1)
        breakdown = {"walk_kernel_ms": float(b[0]) * k, "sort_merge_presample_ms": float(b[1]) * k, "exchange_kernel_ms": float(b[2]) * k,
                     "allreduce_ms": float(b[3]) * k, "what": "max over ranks, per step; allreduce_ms = the collective on the stream it ran on, "
                     "incl. the wait for the slowest rank (blocking exchanges only)",
                     "walk_kernel_ms_min_rank": float(lo[0]) * k, "shard_nnz_min_max": [int(lo[4]), int(b[4])]}

    if rank == 0:
        updates = float(total_nnz) * opt["num_negative_samples"] * steps
        # roofline of the dominant kernel (bpr_update_kernel), HIP events on the backend's stream
        bytes_per_update = 24 * D + 20            # read+write P_u, Q_i, Q_j, 2 biases, key
        kernel_ms = st["kernel_ms"] / max(st["launches"], 1)
        alg_bytes = bytes_per_update * (st["samples"] / max(st["launches"], 1))
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        # PMC counters need rocprofv3: `traffic` is the per-launch figure of the latest separate --pmc passes of this same
        # command (scripts/gpu_profile.sh -> scripts/pmc_summary.py -> profiles/pmc_latest.json), not of this process
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # the library's own rule for the two-triples-per-wave walk (bfh_bpr_set_mode "im_dual"): vdim <= 128 and >= 6144 users per queue
        dual_walk = hog == "3" and knobs.get("im_dual", "-1") != "0" and (knobs.get("im_dual", "-1") == "1" or n_local_users >= 8 * 6144)
        out = {
            "metric": "BPRMF training throughput (interactions/s), ML-20M-shaped synthetic, d=128",
            "value": updates / elapsed, "unit": "updates/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BPRMF sgd, %s-shaped synthetic (%d x %d, %d nnz%s), d=%d, 1 negative/positive, "
                                   "uniform sampling + verify_neg, CSR + factors resident in HBM"
                                   % (args.shape, U, I, nnz, "" if args.scaling == "strong" or world == 1 else " per GPU", D),
                       "parallelism": "1 GPU" if world == 1 else "dp%d: users sharded, Q replicated, %.1f delta all-reduce/epoch (%s)"
                                      % (world, (st["exchanges"] / max(steps, 1)) if comm is not None else args.minibatches,
                                         ("inside the library, transport %s" % os.environ.get("BFH_COMM_TRANSPORT", "RCCL over xGMI")) if comm is not None
                                         else (comm_note or "EMERGENCY PATH through torch.distributed")),
                       "hogwild": {"0": "write-through (sc1) racy stores on item rows", "1": "fp32 atomics on item rows",
                                   "2": "per-XCD item-factor replicas (plain stores through the XCD's L2, merged by the delta rule "
                                        "%.1f times per epoch); popular rows stay chip-wide on fp32 atomics"
                                        % (st["merges"] / max(steps, 1)),
                                   "3": "item-major walk%s: users owned by XCDs (plain stores through the owner's L2), the positive "
                                        "item row in registers with bounded-staleness atomic flushes, negatives in per-XCD replicas "
                                        "merged by the delta rule %.1f times per epoch"
                                        % (", two triples per wave" if dual_walk else "", st["merges"] / max(steps, 1))}[hog]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/pmc_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)",
                         "kernel": ("bpr_item_major_dual_kernel" if dual_walk else "bpr_item_major_kernel") if hog == "3" else "bpr_update_kernel",
                         "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         # SURVEY.md section 8(d)'s stricter variant, reported alongside: one side's row is read and written once per
                         # run of its triples instead of once per triple (16 d + 20 + 8 d / mean degree bytes per update)
                         "strict": {"bytes_per_update": 16 * D + 20 + 8 * D / (nnz / U),
                                    "achieved": achieved * (16 * D + 20 + 8 * D / (nnz / U)) / bytes_per_update,
                                    "frac": achieved * (16 * D + 20 + 8 * D / (nnz / U)) / bytes_per_update / HBM_PEAK_GBS},
                         "launches_per_step": st["launches"] / max(steps, 1),
                         # the same bytes over ALL device time of a step (update launches + replica broadcast / merge kernels)
                         "frac_incl_merge_kernels": (bytes_per_update * st["samples"] / max((st["kernel_ms"] + st["aux_ms"]) * 1e-3, 1e-12)
                                                     / 1e9 / HBM_PEAK_GBS)},
            "epoch_ms": elapsed / steps * 1e3,
        }
        if breakdown is not None:
            out["breakdown"] = breakdown
        if world == 1:
            try:   # the same box's HBM under a trivial kernel, next to the 8 TB/s the fraction is quoted against
                m = measured_stream_bandwidth()
                m["frac_of_triad"] = achieved / m["triad_GBps"] if m["triad_GBps"] > 0 else None
                out["roofline"]["measured_stream"] = m
            except Exception as e:
                out["roofline"]["measured_stream"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(csr)
        if world == 1 and not args.no_extra:
            del obj
            extra = {}
            for name, fn in (("als_ml20m_d128", extra_als), ("warp_ml20m_d256", extra_warp),
                             ("warp_c5_one_gpu", lambda _csr, seed, cpu: extra_warp_c5(seed, cpu=cpu)), ("topk_ml20m_d128_k100", extra_topk),
                             ("sppmi_ml20m_stream_w5", extra_sppmi), ("coo_to_csr_ml20m", extra_ingest)):
                try:
                    extra[name] = fn(csr, args.seed, cpu=not args.no_cpu_baseline)
                except Exception as e:   # the headline line is never lost to a secondary measurement
                    extra[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:   # the top of the reference's own D-sweep (benchmark/README.md:97): d = 160, block 32 -> the wide ALS kernel (T = 5)
                extra["als_ml20m_d160"] = extra_als_wide(csr, args.seed, 160)
            except Exception as e:
                extra["als_ml20m_d160"] = {"error": "%s: %s" % (type(e).__name__, e)}
            out["extra"] = extra
            # the driver's record keeps the scalars of `roofline` / `cpu_baseline` and drops nested objects: configs[2] (ALS) and
            # configs[4] (WARP) at BASELINE size, measured in this process, as flat keys
            rf = out["roofline"]
            ms = rf.get("measured_stream") or {}
            if "triad_GBps" in ms:
                rf["triad_GBps"], rf["frac_of_triad"] = ms["triad_GBps"], ms.get("frac_of_triad")
            a = extra.get("als_ml20m_d128") or {}
            if "epoch_ms" in a:
                rf.update({"als_epoch_ms": a["epoch_ms"], "als_kernel_ms": a["kernel_ms_per_epoch"], "als_hbm_frac": a["hbm"]["frac"],
                           "als_hbm_frac_of_epoch": a["hbm"]["algorithmic_bytes_per_epoch"] / (a["epoch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "als_useful_mfma_frac": a["mfma"]["useful_frac_of_fp32_peak"], "als_issued_mfma_frac_f16": a["mfma"]["frac"]})
            a160 = extra.get("als_ml20m_d160") or {}
            if "epoch_ms" in a160:
                rf.update({"als_d160_epoch_ms": a160["epoch_ms"], "als_d160_kernel_ms": a160["kernel_ms_per_epoch"]})
            w = extra.get("warp_ml20m_d256") or {}
            if "epoch_ms" in w:
                rf.update({"warp_ml20m_epoch_ms": w["epoch_ms"], "warp_ml20m_T": w["mean_scored_negatives_T"],
                           "warp_ml20m_implemented_model_frac": w["implemented_model_frac"]})
            c5 = extra.get("warp_c5_one_gpu") or {}
            if "epoch_ms" in c5:
                last = c5["epochs"][-1]
                rf.update({"warp_c5_epoch_ms": c5["epoch_ms"], "warp_c5_T": c5["mean_scored_negatives_T"], "warp_c5_accepted_frac": last["accepted_frac"],
                           "warp_c5_epochs_run": len(c5["epochs"]), "warp_c5_implemented_model_frac": c5["implemented_model_frac"],
                           "warp_c5_sort_and_gather_ms": last["sort_and_gather_ms"], "warp_c5_trial_kernel_ms": last["trial_kernel_ms"]})
                tr = c5.get("counter_traffic") or {}
                if isinstance(tr, dict) and tr.get("hbm_bytes_per_epoch"):
                    dev_ms = last["trial_kernel_ms"] + last["sort_and_gather_ms"] + last["optimizer_ms"]
                    rf["warp_c5_traffic_frac"] = tr["hbm_bytes_per_epoch"] / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                    rf["warp_c5_traffic_T"] = tr.get("mean_scored_negatives_T")   # the T of the profiled epoch (compare with warp_c5_T)
                    rf["warp_c5_traffic_source"] = "profiles/warp_pmc_latest.json (counter bytes of an earlier run of this workload) over this run's device time"
            cb = out.get("cpu_baseline") or {}
            for name, key in (("als_ml20m_d128", "als"), ("warp_ml20m_d256", "warp_ml20m"), ("warp_c5_one_gpu", "warp_c5")):
                v = (extra.get(name) or {}).get("cpu_baseline") or {}
                if "value" in v and cb:
                    cb["%s_value" % key], cb["%s_unit" % key] = v["value"], v["unit"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.all_reduce(torch.zeros(1))
        del obj
        comm = None
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
