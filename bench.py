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
CPU_WHAT_SHORT = "oracle/buffalo_oracle.cc: restatement of the reference CPU path, the reference's flags (-O3 -fopenmp -mavx2 -mfma)"
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


def cpu_baseline(csr, target_seconds=12.0, all_cores=False):
    """The oracle (restatement of the reference CPU path, reference's compile flags) timed on this
    box's host cores over a bounded prefix of the same workload.  `all_cores` (--cpu-all-cores): additionally one worker per host core -- on the
    256-core GPU box that is the reference's contended job queue at HALF the 8-worker rate for 6 s of the default run; measured once
    (BENCH_r05.json: 3.38 M/s against 7.40 M/s), off by default since round 6."""
    from oracle import oracle as orc
    from buffalo_amd import synth
    orc.build()
    cores = os.cpu_count() or 1
    I = csr.num_items

    def run(n_users, workers=cores, cls=None):
        nnz = int(csr.indptr[n_users - 1])
        opt = bpr_options(1, accelerator=False, num_workers=workers)
        P, Q, Qb = synth.init_factors(n_users, I, D, seed=7)
        o = (cls or orc.OracleBPRMF)()
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
    # the reference's own benchmark setting is 8 workers (tests/algo/test_performance.py:53): same port, sized by its own probe
    # (the job queue of the CPU path is contended, so fewer workers can be FASTER than all cores)
    nnz_p, dt_p = run(probe_users, workers=8)
    want8 = int(min(csr.nnz, max(nnz_p, nnz_p / dt_p * 6.0)))
    nnz8, dt8 = run(min(csr.num_users, int(np.searchsorted(csr.indptr, want8)) + 1), workers=8)
    # the headline figure is the 8-worker one -- the reference's own benchmark setting, and the faster of the two (its job queue
    # is contended: all cores are SLOWER); the all-core run is kept beside it
    out = {"value": nnz8 / dt8, "unit": "updates/s", "cores": 8, "kind": CPU_KIND, "what": CPU_WHAT,
           "sample": "first %d interactions (1 epoch) of the same matrix, 8 std::thread workers (the reference's benchmark setting, "
                     "tests/algo/test_performance.py:53), %.1f s" % (nnz8, dt8)}
    if all_cores:
        nnz0, dt0 = run(probe_users)
        want = int(min(csr.nnz, max(nnz0, nnz0 / dt0 * target_seconds)))
        n_users = min(csr.num_users, int(np.searchsorted(csr.indptr, want)) + 1)
        nnz1, dt1 = run(n_users)
        out.update({"value_all_cores": nnz1 / dt1, "all_cores": cores,
                    "sample_all_cores": "first %d users (%d interactions, 1 epoch), %d workers, %.1f s" % (n_users, nnz1, cores, dt1)})
    # the bracket: the reference's OWN algo.cc / bpr.cc compiled unmodified on stand-ins for Eigen / json11 / spdlog (oracle/_ref, built in
    # the build container by __graft_entry__.build(); NOT the reference binary -- an Eigen expression evaluates as the stand-in reads it),
    # same sample, same 8 workers
    try:
        from oracle import ref_sgd
        if os.path.exists(ref_sgd._path("bpr")):
            n8 = min(csr.num_users, int(np.searchsorted(csr.indptr, want8)) + 1)
            nnz_r, dt_r = run(n8, workers=8, cls=ref_sgd.RefBPRMF)
            out.update({"reference_on_stand_ins_value": nnz_r / dt_r, "reference_on_stand_ins_kind": "reference-on-stand-ins",
                        "reference_on_stand_ins_sample": "lib/algo.cc + bpr.cc compiled unmodified on oracle/stand_in_3rd, first %d interactions, "
                                                         "8 workers, %.1f s" % (nnz_r, dt_r)})
    except Exception as e:
        out["reference_on_stand_ins_error"] = "%s: %s" % (type(e).__name__, e)
    return out


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
    out = {"config": "ALS iALS++ (block 32, 3 CG steps), ml20m-shaped synthetic (%d x %d, %d nnz, values 1+Poisson(1)), d=%d, Gramian in split-f16 "
                     "(two f16 pieces per factor, ~22-bit products, fp32 accumulate; everything else f32), rowwise + colwise CSR and factors "
                     "resident in HBM" % (U, I, nnz, D),
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
    if cpu:
        try:
            out["parity"] = als_parity_block(g, P, Q, csr, vals, col, epoch)
        except Exception as e:
            out["parity"] = {"error": "%s: %s" % (type(e).__name__, e)}
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


def als_parity_block(g, P, Q, csr, vals, col, epoch):
    """configs[2] against the REFERENCE PATH in absolute numbers, in the bench line (the statement of
    tests/test_als_gpu.py::test_config3_warm_epoch_matches_the_oracle_path, without the float64 envelopes): the model is warm (the timed
    epochs), that state goes to the device handle and to the oracle, each runs ONE epoch, and the line carries
      user_max / item_max : max |x_hip - x_oracle| over all rows, relative to the largest entry of the oracle's factor;
      item_max_reorder    : the same distance between the oracle and ITSELF with every item row's entries in reverse order (a legal
                            reordering of its fp32 sums; the item systems are ill-conditioned even warm) -- the yardstick for item_max;
      top10_overlap(_reorder): mean overlap of 2,000 sampled users' top-10 lists, HIP model vs oracle model (oracle vs reordered oracle)."""
    from oracle import oracle as orc
    orc.build()
    U, I = csr.num_users, csr.num_items
    g.synchronize(True)
    Po, Qo = P.copy(), Q.copy()
    o = orc.OracleALS()
    path = _opt_file(dict(ALS_OPT, accelerator=False, num_workers=os.cpu_count() or 16))
    assert o.init(path)
    os.unlink(path)
    o.initialize_model(Po, Qo)
    t0 = time.perf_counter()
    o.precompute(0)
    o.partial_update(0, U, csr.indptr, csr.keys, vals, 0)
    Po_mid, Qw = Po.copy(), Qo.copy()
    o.precompute(1)
    o.partial_update(0, I, col["indptr"], col["key"], col["val"], 1)
    ip = col["indptr"]
    starts = np.concatenate([[0], ip[:-1]])
    rid = np.repeat(np.arange(I), np.diff(np.concatenate([[0], ip])))
    rev = (starts[rid] + (ip[rid] - 1 - np.arange(len(rid), dtype=np.int64))).astype(np.int64)
    Pr, Qr = Po_mid.copy(), Qw.copy()
    o2 = orc.OracleALS()
    path = _opt_file(dict(ALS_OPT, accelerator=False, num_workers=os.cpu_count() or 16))
    assert o2.init(path)
    os.unlink(path)
    o2.initialize_model(Pr, Qr)
    o2.precompute(1)
    o2.partial_update(0, I, ip, np.ascontiguousarray(col["key"][rev]), np.ascontiguousarray(col["val"][rev]), 1)
    cpu_s = time.perf_counter() - t0
    epoch()                                          # the device handle, one epoch from the same warm state
    g.synchronize(True)

    def dist(X, Xo):
        return float(np.abs(X - Xo).max() / max(float(np.abs(Xo).max()), 1e-30))
    users = np.random.default_rng(11).choice(U, 2000, replace=False)

    def top10(Pm, Qm):
        return np.argsort(-(Pm[users] @ Qm.T), axis=1)[:, :10]

    def overlap(a, b):
        return float(np.mean([len(set(x) & set(y)) / 10.0 for x, y in zip(a, b)]))
    to = top10(Po, Qo)
    return {"user_max": dist(P, Po), "item_max": dist(Q, Qo), "item_max_reorder": dist(Qr, Qo),
            "top10_overlap": overlap(top10(P, Q), to), "top10_overlap_reorder": overlap(top10(Po_mid, Qr), to),
            "what": "one epoch from the same warm state, HIP vs the oracle (restatement of als.cc:211-358) over ALL rows; *_reorder = the oracle "
                    "against itself with the item rows' entries reversed; oracle time %.1f s" % cpu_s}


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
            "accepted_frac": acc / nnz, "algorithmic_bytes": alg, "algorithmic_GBps": alg / dev_s / 1e9,
            # SURVEY 8(d)'s formula counts six gradient-row read-modify-writes per positive that the gather-by-item formulation never performs:
            # this ratio can exceed 1 and is NOT a bandwidth fraction (that is counter_traffic / implemented_model_frac)
            "survey_formula_bytes_over_peak": alg / dev_s / 1e9 / HBM_PEAK_GBS,
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


def _warp_epochs(g, U, indptr, nnz, d, I, epochs, until_T=None, max_epochs=None, after_first_jobs=None):
    """`epochs` epochs; with `until_T`, further ones (at most `max_epochs` in all) until the trial loop scores that many negatives
    per positive -- the regime training lives in, not the first epochs' T = 1 where every first draw violates the margin.
    `after_first_jobs`: called once between the first epoch's add_jobs and its update_parameters (the results check reads the accumulated
    gradient rows there; the copy sits in epoch 1's time, which is never the quoted one)."""
    eps = []
    e = 0
    while e < epochs or (until_T is not None and e < (max_epochs or epochs) and eps[-1]["mean_scored_negatives_T"] < until_T):
        g.reset_stats()
        t0 = time.perf_counter()
        g.add_jobs(0, U, indptr, None)
        if e == 0 and after_first_jobs is not None:
            after_first_jobs()
        g.update_parameters()
        dt = time.perf_counter() - t0
        eps.append(warp_epoch_row(g.stats(), nnz, d, U, I, dt))
        e += 1
    return eps


def warp_results_check(indptr, keys, P0, Q0, Qb0, grad_device_full, n_users, total_nnz, opt):
    """Results at the size the device ran (/root/reference/lib/algo_impl/warp/warp.cc:128-158).  P and Q are frozen inside a WARP epoch
    (warp.cc:157-159 accumulates gradients, the optimizer step comes with update_parameters), so the gradient row a user has accumulated at the end of the
    FIRST epoch's add_jobs depends on its own positives alone: the oracle (counter sampler -- the device's draw function --, csr order, one inline worker:
    deterministic) run on the first `n_users` users from the same initial state reproduces what the full-size device epoch did to them.  Compared:
      * the accumulated gradient rows of those users, full-size device run vs oracle (max |diff| / max |row|, bound 1e-4);
      * against a second, small device run on the same users: the number of scored negatives (trials) and of accepted positives -- IDENTICAL -- and its
        gradient rows (1e-4);
      * the users' rows after the optimizer step, reported as the share of coordinates further than 1e-4 apart: adagrad's first step is lr * sign(g), so a
        coordinate whose 100 contributions cancel to rounding noise may step the other way (bound 1e-5 of the coordinates; the gradients are the sharp part)."""
    from buffalo_amd.backend import CyWARP
    from oracle import oracle as orc
    orc.build()
    n = int(n_users)
    m = int(indptr[n - 1])
    ip, k = np.ascontiguousarray(indptr[:n]), np.ascontiguousarray(keys[:m])
    I, d = Q0.shape
    Po = np.ascontiguousarray(P0[:n]).copy()
    o = orc.OracleWARP()
    path = _opt_file(dict(opt, accelerator=False, num_workers=1))
    assert o.init(path)
    os.unlink(path)
    o.initialize_model(Po, Q0.copy(), Qb0.copy(), total_nnz)
    o.set_cumulative_table(np.zeros(I, np.int64), I)
    o.set_modes(sampler="counter", pos_order="csr", inline=True)
    o.launch_workers()
    t0 = time.perf_counter()
    o.add_jobs(0, n, ip, k)
    g_or = o.state("gradP").reshape(-1, d)[:n].copy()
    so = o.stats()
    cpu_s = time.perf_counter() - t0
    o.update_parameters()
    o.join()
    # the same users through a small device run: the counters of the full-size run are totals over ALL users
    Ps = np.ascontiguousarray(P0[:n]).copy()
    sub = CyWARP()
    path = _opt_file(dict(opt))
    assert sub.init(path)
    os.unlink(path)
    sub.initialize_model(Ps, Q0.copy(), Qb0.copy(), total_nnz, True)
    sub.set_cumulative_table(np.zeros(I, np.int64), I)
    sub.set_resident_csr(ip, k)
    sub.add_jobs(0, n, ip, None)
    g_sub = sub.device_tensor("gradP", (n, d)).cpu().numpy().copy()
    ss = sub.stats()
    sub.update_parameters()
    sub.synchronize(True)
    del sub
    gden = float(np.abs(g_or).max()) or 1.0
    e_full = float(np.abs(grad_device_full - g_or).max() / gden)
    e_sub = float(np.abs(g_sub - g_or).max() / gden)
    pden = float(np.abs(Po).max()) or 1.0
    far = float((np.abs(Ps - Po) > 1e-4 * pden).mean())
    counts_o = (int(so["scored_negatives"]), int(so["updates"]))
    counts_d = (int(ss["scored_negatives"]), int(ss["accepted"]))
    return {"users": n, "positives": m, "oracle_seconds": cpu_s,
            "scored_negatives_oracle": counts_o[0], "accepted_oracle": counts_o[1], "scored_negatives_device": counts_d[0], "accepted_device": counts_d[1],
            "grad_rows_full_run_vs_oracle": e_full, "grad_rows_sample_run_vs_oracle": e_sub, "grad_rows_max_abs": gden,
            "rows_after_step_share_of_coordinates_apart": far,
            "agrees_with_device": bool(counts_o == counts_d and e_full < 1e-4 and e_sub < 1e-4 and gden > 1e-3 and far < 1e-5),
            "what": "first epoch, first %d users: oracle (counter sampler, one inline worker) vs the full-size device epoch's accumulated gradient rows of those users "
                    "(max |diff| / max |g|, bound 1e-4) and vs a device run on those users alone (trial and accept counts, identical)" % n}


def _warp_summary(eps, out):
    """The epoch to quote is the LAST one: epochs 1-2 start from near-zero factors where every first negative violates the margin
    (T = 1, everything accepted -- the easy regime); by the third the trial loop rejects (T ~ 4 on the ML-20M shape)."""
    last = eps[-1]
    out.update({"epochs": eps, "quoted_epoch": len(eps) - 1, "epoch_ms": last["epoch_ms"], "positives_per_s": last["positives_per_s"],
                "mean_scored_negatives_T": last["mean_scored_negatives_T"], "algorithmic_GBps": last["algorithmic_GBps"], "survey_formula_bytes_over_peak": last["survey_formula_bytes_over_peak"],
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


def warp_c5_inputs(skew=True, u0=0, u1=None, users=10_000_000, items=1_000_000):
    """BASELINE configs[4]'s shape for ONE GPU: 10 M users x 1 M items, 1 B interactions, d=256.  Every user has 100 items, one in
    each 10,000-wide band of the catalogue (sorted keys, no duplicates).  `skew`: the item inside a band is floor(10^4 x^3) for a
    hashed x in [0, 1) -- a popularity law (the head item of a band is chosen by 4.6 % of the users), so that there is something to
    rank and the trial loop leaves the T = 1 regime as it does on real data; without it (round 2's generator) every epoch accepts
    the first draw.  Factors signed N(0, 1/d^2) (Q-18), P tiled from 65,536 distinct rows to keep host generation to seconds.
    41.9 GB resident (P, Q, gradients, adagrad state, keys, row ids)."""
    I5, deg, d = items, 100, WARP_D
    u1 = users if u1 is None else u1
    U5 = u1 - u0                                   # this rank's users [u0, u1) of `users` (N > 1: --workload warp_c5)
    step = I5 // deg
    keys = np.empty((U5, deg), dtype=np.int32)
    band_hash = ((np.arange(deg, dtype=np.int64) * 104729) % step).astype(np.int32)
    band_base = np.arange(deg, dtype=np.int32) * step
    for c0 in range(0, U5, 500_000):              # int32 throughout: ~11 s for the 10^9 keys on the host
        u = np.arange(u0 + c0, u0 + min(U5, c0 + 500_000), dtype=np.int64)
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
        keys[c0:c0 + u.shape[0]] = off
    keys = keys.reshape(-1)
    indptr = (np.arange(U5, dtype=np.int64) + 1) * deg
    rng = np.random.default_rng(7)
    base = (rng.normal(size=(65536, d)) / d).astype(np.float32)
    P = np.ascontiguousarray(base[np.arange(u0, u1, dtype=np.int64) % 65536])
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
    chk = {}
    n_chk = int(os.environ.get("BFH_WARP_CHECK_USERS", "20000"))

    def grab():   # the accumulated gradient rows of the first users after epoch 1's add_jobs, straight from HBM
        chk["G"] = g.device_tensor("gradP", (U, d))[:n_chk].cpu().numpy().copy()
    P0 = np.ascontiguousarray(P[:n_chk]).copy() if cpu else None
    Q0 = Q.copy() if cpu else None
    eps = _warp_epochs(g, U, indptr, nnz, d, I, epochs, until_T=3.0, max_epochs=12, after_first_jobs=grab if cpu else None)
    out = _warp_summary(eps, {
        "config": "WARP adagrad, dot score, max_trials 500, configs[4]-shaped synthetic (%d x %d, %d nnz), d=%d, f32, ONE GPU, everything "
                  "resident in HBM" % (U, I, nnz, d),
        "host_generation_s": gen_s, "upload_s": up_s,
        "hbm_resident_GB": (3 * (U + I) * d * 4 + keys.nbytes * 2 + indptr.nbytes) / 1e9})
    tr = _warp_counter_traffic("c5")
    if tr:
        out["counter_traffic"] = tr
    # ---- the SEARCHING regime at this size: with the option default margin (1.0, rows inside the unit ball) every positive of this
    # synthetic shape finds a violator within two counted trials (T plateaus at 1.8, accepted_frac 1.0): the easy regime.  The trained model is taken
    # back, the margin set to the 30 % quantile of x_ui - x_uj over 200,000 sampled triples (about 3 of 10 draws violate: T ~ 1 / 0.3 counted trials, the
    # construction of tests/test_warp_scale_gpu.py), and three more epochs run from there with a fresh handle: trial kernel at T >= 3 on 10^9 positives.
    try:
        g.synchronize(True)                      # P, Q, Qb <- the device's model (through the pinned ring)
        del g
        rs = np.random.default_rng(1)
        su = rs.integers(0, U, 200000)
        si = keys[su.astype(np.int64) * 100 + rs.integers(0, 100, 200000)]
        sj = rs.integers(0, I, 200000)
        diff = np.einsum("ij,ij->i", P[su], Q[si] - Q[sj])
        thr = float(np.quantile(diff, 0.30))
        g2 = CyWARP()
        # (lr 1e-6: a fresh adagrad state takes lr * sign(g) as its first step -- at the option's 0.05 that would throw the trained rows away; the work per
        #  epoch does not depend on the step size)
        path = _opt_file(dict(WARP_OPT, threshold=thr, lr=1e-6, min_lr=1e-6))
        assert g2.init(path)
        os.unlink(path)
        g2.sync_every_epoch = False
        g2.initialize_model(P, Q, Qb, nnz, True)
        g2.set_resident_csr(indptr, keys)
        eps2 = _warp_epochs(g2, U, indptr, nnz, d, I, 3)
        del g2
        last2 = eps2[-1]
        out["searching_regime"] = {"threshold": thr, "what": "margin = the 30 % quantile of x_ui - x_uj over 200,000 sampled triples of the trained model; three epochs "
                                                              "from that model, the last one quoted", "epochs": eps2, "epoch_ms": last2["epoch_ms"],
                                   "mean_scored_negatives_T": last2["mean_scored_negatives_T"], "accepted_frac": last2["accepted_frac"],
                                   "trial_kernel_ms": last2["trial_kernel_ms"], "sort_and_gather_ms": last2["sort_and_gather_ms"],
                                   "implemented_model_frac": last2["implemented_model_frac"]}
    except Exception as e:  # noqa: BLE001
        out["searching_regime"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if cpu:
        n100 = U // 100
        out["cpu_baseline"] = _warp_cpu_baseline(indptr[:n100], keys[:int(indptr[n100 - 1])], I, d, seed, seconds=4.0)
        out["cpu_baseline"]["sample"] += "; the first 1/100 of the users against the full item table, to be scaled linearly (SURVEY 8(d))"
        try:
            out["results_check"] = warp_results_check(indptr, keys, P0, Q0, np.zeros((I, 1), np.float32), chk["G"], n_chk, nnz, WARP_OPT)
            out["agrees_with_device"] = out["results_check"]["agrees_with_device"]
        except Exception as e:  # noqa: BLE001 -- reported, never silently dropped
            out["results_check"] = {"error": "%s: %s" % (type(e).__name__, e)}
            out["agrees_with_device"] = False
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


def extra_text_ingest(csr, seed, cpu=True, lines=2_000_000):
    """SURVEY.md section 8 f.2, the step BEFORE the arrays exist: the working text file of buffalo's data creation ("row col val" lines, 1-based,
    data/mm.py:175-234) -> rowwise group, parse included (bfh_text_to_csr = fileio.hpp:263-420 on the device), beside the reference's own
    compiled builder on the SAME file.  A 2 M-line sample of the matrix's records in shuffled order (writing 20 M lines of text from Python
    would take longer than the rest of this run)."""
    import tempfile
    from buffalo_amd import ingest
    rng = np.random.default_rng(seed)
    n = min(lines, csr.nnz)
    pick = rng.choice(csr.nnz, n, replace=False)
    rows, cols = csr.rows()[pick].astype(np.int64) + 1, csr.keys[pick].astype(np.int64) + 1
    stars = (1 + (pick % 10)) * 0.5                                   # half-star ratings 0.5 .. 5.0
    text = ("\n".join("%d %d %s" % t for t in zip(rows.tolist(), cols.tolist(), stars.tolist())) + "\n").encode()
    ingest.text_to_csr(text, n, csr.num_users, csr.num_items, 1)
    t0 = time.perf_counter()
    g, st = ingest.text_to_csr(text, n, csr.num_users, csr.num_items, 1, with_stats=True)
    dt = time.perf_counter() - t0
    want = ingest.coo_to_csr((rows - 1).astype(np.int32), (cols - 1).astype(np.int32), stars.astype(np.float32), csr.num_users, csr.num_items)
    out = {"config": "%d text lines (%d bytes) 'row col val' -> rowwise group of %d x %d: host bytes in -> host (indptr, key, val) out" % (n, len(text), csr.num_users, csr.num_items),
           "lines": n, "bytes": len(text), "device_ms_incl_upload": st["kernel_ms"], "wall_ms": dt * 1e3, "lines_per_s_wall": n / dt,
           "lines_reparsed_on_host": st["merges"],
           "identical_to_the_array_path": bool(np.array_equal(g["indptr"], want["indptr"]) and np.array_equal(g["key"], want["key"]) and np.array_equal(g["val"], want["val"]))}
    if cpu:
        try:
            from oracle import ref_fileio as rf
            if os.path.exists(rf._LIB_PATH):
                workers = os.cpu_count() or 1
                with tempfile.TemporaryDirectory() as d:
                    src = os.path.join(d, "working.txt")
                    with open(src, "wb") as f:
                        f.write(text)
                    t0 = time.perf_counter()
                    nf = rf.lib().ref_sort_and_compressed_binarization(src.encode(), d.encode(), n, csr.num_users, 1, workers)
                    dr = time.perf_counter() - t0
                    assert nf == workers + 1
                    ok = bool(np.array_equal(np.fromfile(os.path.join(d, "indptr.bin"), dtype=np.int64), g["indptr"]))
                out["cpu_baseline"] = {"value": n / dr, "unit": "lines/s", "cores": workers, "kind": "reference",
                                       "what": "the reference's own buffalo/data/fileio.hpp compiled from its source (oracle/_ref): "
                                               "_sort_and_compressed_binarization on the same file (parse, stable sort, indptr, binary chunks)",
                                       "sample": "%d lines, %d workers, %.2f s" % (n, workers, dr), "same_indptr_as_device": ok}
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


# ------------------------------------------------------------------------------------------------------------------------------
# The line.  The driver parses the LAST stdout line and keeps an 8 KB tail of stdout (round 4's 27 KB line was cut in the middle and
# the round went unrecorded), so the last line is held under LINE_LIMIT: the contract's head keys, `config`, `roofline` and
# `cpu_baseline` as FLAT scalars.  Everything else (per-epoch lists, byte models, per-kernel tables, the long descriptions) goes to
# `bench_extra.json` (repo root and gpurun_out/) and to an EARLIER stdout line prefixed "BENCH_EXTRA ".
LINE_LIMIT = 4096
REFBENCH_QB_REFERENCE = 183.0   # |Qb| of the reference path (threaded oracle) after lr 0.05 -> 0.0001 over 10 epochs on load_matrix("ml20m", 7), init seed 7
HEAD_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
# flat roofline keys the line can do without, first to go first (the contract's own keys are never dropped)
ROOFLINE_OPTIONAL = ("warp_c5_traffic_source", "traffic_source", "warp_c5_sort_and_gather_ms", "warp_c5_trial_kernel_ms", "warp_c5_epochs_run",
                     "warp_ml20m_implemented_model_frac", "warp_c5_implemented_model_frac", "strict_frac", "strict_bytes_per_update",
                     "frac_incl_merge_kernels", "launches_per_step", "als_issued_mfma_frac_f16", "als_d160_kernel_ms", "copy_GBps")


def _short(v, n=200):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 3] + "..."


def _num(v):
    if isinstance(v, bool) or not isinstance(v, float):
        return v
    if v != v or v in (float("inf"), float("-inf")):
        return None
    return float("%.6g" % v)


def _flat(d, strlen=200):
    """The scalar members of a dict (numbers rounded to 6 digits, strings cut): what the driver's record keeps."""
    return {k: _num(_short(v, strlen)) for k, v in d.items() if v is None or isinstance(v, (bool, int, float, str))}


def compact_line(out, extra_file="bench_extra.json", limit=LINE_LIMIT):
    """The LAST stdout line from the full result: <= `limit` bytes, json.loads-able, head + config + flat roofline + flat cpu_baseline."""
    line = {k: _num(out.get(k)) for k in HEAD_KEYS}
    line["config"] = _flat(out.get("config") or {}, 360)
    rf = _flat(out.get("roofline") or {})
    st = (out.get("roofline") or {}).get("strict") or {}
    if "frac" in st:
        rf["strict_frac"], rf["strict_bytes_per_update"] = _num(st["frac"]), _num(st["bytes_per_update"])
    line["roofline"] = rf
    if out.get("cpu_baseline"):
        line["cpu_baseline"] = _flat(dict(out["cpu_baseline"], what=CPU_WHAT_SHORT if out["cpu_baseline"].get("kind") == "port" else
                                          out["cpu_baseline"].get("what", "")), 220)
    if out.get("breakdown"):
        line["breakdown"] = _flat(out["breakdown"], 120)
    for k in ("rccl_ranks", "transport"):             # N > 1: what the live communicator reports (bfh_comm_size / bfh_comm_transport)
        if k in out:
            line[k] = out[k]
    line["extra_file"] = extra_file
    s = json.dumps(line)
    for k in ROOFLINE_OPTIONAL:                       # never needed at today's size; the guard that keeps the driver's record whole
        if len(s) <= limit:
            break
        line["roofline"].pop(k, None)
        s = json.dumps(line)
    if len(s) > limit:
        for blk in ("config", "cpu_baseline", "breakdown"):
            for k, v in list((line.get(blk) or {}).items()):
                if isinstance(v, str):
                    line[blk][k] = _short(v, 80)
        s = json.dumps(line)
    assert len(s) <= limit, "bench line is %d bytes (> %d)" % (len(s), limit)
    return s


def emit(out):
    """Side file + the two stdout lines (extras first, the driver's line LAST)."""
    extra = {k: v for k, v in out.items() if k not in HEAD_KEYS}
    blob = json.dumps(dict({k: out.get(k) for k in HEAD_KEYS}, **extra))
    for path in (os.path.join(ROOT, "bench_extra.json"), os.path.join(ROOT, "gpurun_out", "bench_extra.json")):
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                f.write(blob)
        except OSError:
            pass
    sys.stdout.write("BENCH_EXTRA " + blob + "\n")
    sys.stdout.write(compact_line(out) + "\n")
    sys.stdout.flush()


def counter_traffic(name="pmc_latest.json", key="hbm_bytes_per_launch"):
    """Counter bytes of the separate rocprofv3 --pmc passes (scripts/gpu_profile.sh -> scripts/pmc_summary.py).  The file is from an
    EARLIER process: it is only used when it was taken on the kernel sources this process runs (`csrc_sha16` stamped by the summary
    script = buffalo_amd._build.source_fingerprint() now); otherwise traffic is null and the reason is in `traffic_source`."""
    from buffalo_amd import _build
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, "profiles/%s absent" % name
    now = _build.source_fingerprint()
    if d.get("csrc_sha16") != now:
        return None, "profiles/%s was taken on other kernel sources (%s, now %s): not used" % (name, d.get("csrc_sha16"), now)
    return d.get(key), "profiles/%s (rocprofv3 --pmc passes of this command on these sources, csrc %s)" % (name, now)


class Ctx:
    """One rank's view of the job: torch.distributed is the CONTROL plane only (rendezvous, barrier, max of the elapsed times -- gloo on
    CPU tensors); the data plane is the library's own communicator (bfh_comm_*: RCCL over xGMI)."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
        assert self.world == args.gpus, "WORLD_SIZE (%d) != --gpus (%d)" % (self.world, args.gpus)
        # test hooks (1-GPU boxes): BFH_DEVICE_OVERRIDE pins every rank to one device; BFH_COMM_TRANSPORT=shm selects the library's
        # shared-memory test transport (RCCL refuses two ranks on one GPU) and with it gloo as the control plane
        if "BFH_DEVICE_OVERRIDE" in os.environ:
            self.local_rank = int(os.environ["BFH_DEVICE_OVERRIDE"])
        torch.cuda.set_device(self.local_rank)
        self.dist = None
        self.comm = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            backend = os.environ.get("BFH_DIST_BACKEND", "gloo" if os.environ.get("BFH_COMM_TRANSPORT") == "shm" else "nccl")
            dist.init_process_group("cpu:gloo,cuda:nccl" if backend == "nccl" else backend)
            self.dist = dist

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.all_reduce(self.torch.zeros(1))                # CPU tensor: a gloo barrier that never touches the data plane
        self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.dist is None:
            return float(x)
        t = self.torch.tensor([float(x)], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def minmax(self, vals):
        t = self.torch.tensor([float(v) for v in vals], dtype=self.torch.float64)
        lo = t.clone()
        if self.dist is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN)
        return [float(v) for v in lo], [float(v) for v in t]

    def make_comm(self):
        """The library communicator of this rank.  There is NO other data plane: a node on which it cannot be built fails the run on every
        rank (a line the library did not produce is worth nothing) -- the reason goes to stderr."""
        from buffalo_amd.backend import Comm
        torch, dist = self.torch, self.dist
        ok, note, comm = 1, None, None
        try:
            uid = torch.frombuffer(bytearray(Comm.unique_id() if self.rank == 0 else bytes(128)), dtype=torch.uint8).clone()
            dist.broadcast(uid, src=0)                           # CPU tensor -> gloo
            comm = Comm(self.world, self.rank, bytes(uid.numpy().tobytes()), self.local_rank)
            comm.self_test()
            assert comm.size() == self.world, "the communicator reports %d ranks, WORLD_SIZE is %d" % (comm.size(), self.world)
        except Exception as e:
            ok, note = 0, "%s: %s" % (type(e).__name__, e)
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)               # every rank takes the same decision
        if int(flag.item()) != 1:
            raise RuntimeError("rank %d: the library communicator (bfh_comm_create: RCCL) could not be built on every rank -- %s"
                               % (self.rank, note or "another rank failed"))
        self.comm = comm
        return comm

    def transport(self):
        """What the live communicator says it runs on ("rccl 2.x.y"; the one-GPU rehearsal: "shm-test")."""
        return self.comm.transport() if self.comm is not None else "none"

    def comm_keys(self):
        """Top-level scalars of every N > 1 line: the proof that the data plane saw N ranks (bfh_comm_size = ncclCommCount of the live communicator)."""
        if self.comm is None:
            return {}
        return {"rccl_ranks": self.comm.size(), "transport": self.comm.transport()}

    def close(self):
        if self.dist is not None:
            self.dist.all_reduce(self.torch.zeros(1))
            self.dist.destroy_process_group()


def timed_steps(ctx, step, steps, warmup, before_timing=None, after_steps=None):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides; the MAX over ranks."""
    for _ in range(warmup):
        step()
    if before_timing is not None:
        before_timing()
    ctx.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if after_steps is not None:
        after_steps()
    ctx.barrier()
    return ctx.max_over_ranks(time.perf_counter() - t0)


def extra_bpr_lr005(csr, seed, epochs=10):
    """The reference's own BPRMF benchmark setting (/root/reference/benchmark/models.py:86-93): lr 0.05 decaying to min_lr 0.0001 over
    num_iters 10, everything else the option defaults -- 10 epochs of the same walk on the same matrix, per-launch kernel time."""
    from buffalo_amd import synth
    from buffalo_amd.backend import CyBPR
    U, I, nnz = csr.num_users, csr.num_items, csr.nnz
    P, Q, Qb = synth.init_factors(U, I, D, seed=seed)
    g = CyBPR()
    path = write_opt(bpr_options(num_iters=epochs, seed=seed, lr=0.05, min_lr=0.0001))
    assert g.init(path)
    os.unlink(path)
    g.sync_every_epoch = False
    g.initialize_model(P, Q, Qb, nnz, True)
    g.set_cumulative_table(np.zeros(I, np.int64), I)
    g.set_resident_csr(csr.indptr, csr.keys)
    per = []
    t_all = time.perf_counter()
    for _ in range(epochs):
        g.reset_stats()
        t0 = time.perf_counter()
        g.add_jobs(0, U, csr.indptr, None)
        g.update_parameters()
        import torch
        torch.cuda.synchronize()
        st = g.stats()
        per.append({"epoch_ms": (time.perf_counter() - t0) * 1e3, "kernel_ms_per_launch": st["kernel_ms"] / max(st["launches"], 1),
                    "launches": st["launches"], "aux_ms": st["aux_ms"]})
    wall = time.perf_counter() - t_all
    g.synchronize(True)
    bytes_per_launch = (24 * D + 20) * nnz / max(per[-1]["launches"], 1)
    # epoch 1 carries the one-off plan build (sort of the entries by queue / block / item); the kernel figure is over epochs 2..
    k = [p["kernel_ms_per_launch"] for p in per[1:]] or [per[0]["kernel_ms_per_launch"]]
    kernel_ms = sum(k) / len(k)
    out = {"config": "BPRMF sgd, lr 0.05 -> min_lr 0.0001 over %d epochs (benchmark/models.py:86-93), ml20m-shaped synthetic, d=%d" % (epochs, D),
           "epochs": per, "kernel_ms_per_launch": kernel_ms, "kernel_ms_per_launch_first": per[0]["kernel_ms_per_launch"],
           "kernel_ms_per_launch_max": max(p["kernel_ms_per_launch"] for p in per),
           "frac": bytes_per_launch / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "epoch_ms_after_first": sum(p["epoch_ms"] for p in per[1:]) / max(len(per) - 1, 1),
           "updates_per_s_after_first": nnz * max(len(per) - 1, 1) / max(sum(p["epoch_ms"] for p in per[1:]) * 1e-3, 1e-12),
           "wall_s": wall, "finite": bool(np.isfinite(P).all() and np.isfinite(Q).all() and np.isfinite(Qb).all()),
           "norm_P": float(np.linalg.norm(P)), "norm_Q": float(np.linalg.norm(Q)), "norm_Qb": float(np.linalg.norm(Qb))}
    del g
    return out


def run_bpr(args, ctx):
    """The headline: BASELINE configs[1] (N = 1) / configs[3] (N > 1) -- BPRMF sgd, ML-20M shape, d = 128."""
    torch, dist = ctx.torch, ctx.dist
    from buffalo_amd import synth
    from buffalo_amd.backend import CyBPR
    from buffalo_amd.dist import shard_csr
    world, rank, local_rank = ctx.world, ctx.rank, ctx.local_rank

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
    if args.lr is not None:
        opt["lr"] = args.lr
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
    comm = None
    if world > 1:
        comm = ctx.make_comm()                 # raises on every rank when RCCL is unavailable: no torch.distributed data path
        obj.set_comm(comm)
    edges = np.linspace(0, n_local_users, (args.minibatches if world > 1 else 1) + 1).astype(int)

    def step():
        for a, b in zip(edges[:-1], edges[1:]):
            obj.add_jobs(int(a), int(b), ip, None)          # an empty chunk still enters the call's collectives
        obj.update_parameters()

    # the exchange still in flight belongs to the timed region
    elapsed = timed_steps(ctx, step, steps, warmup, before_timing=obj.reset_stats,
                          after_steps=(obj.comm_flush if comm is not None else None))
    st = obj.stats()
    breakdown = None
    if world > 1:
        # per step, the slowest rank of each: where a scaling run's time goes (device times from HIP events inside the library)
        lo, hi = ctx.minmax([st["kernel_ms"], st["aux_ms"], st["exchange_kernel_ms"], st["allreduce_ms"], float(local_nnz)])
        k = 1.0 / max(steps, 1)
        breakdown = {"walk_kernel_ms": hi[0] * k, "sort_merge_presample_ms": hi[1] * k, "exchange_kernel_ms": hi[2] * k,
                     "allreduce_ms": hi[3] * k, "what": "max over ranks, per step; allreduce_ms = the collective on the stream it ran on, "
                     "incl. the wait for the slowest rank (blocking exchanges only)",
                     "walk_kernel_ms_min_rank": lo[0] * k, "shard_nnz_min": int(lo[4]), "shard_nnz_max": int(hi[4])}
    if rank != 0:
        del obj
        return None

    updates = float(total_nnz) * opt["num_negative_samples"] * steps
    # roofline of the dominant kernel, HIP events on the backend's stream
    bytes_per_update = 24 * D + 20            # read+write P_u, Q_i, Q_j, 2 biases, key
    kernel_ms = st["kernel_ms"] / max(st["launches"], 1)
    alg_bytes = bytes_per_update * (st["samples"] / max(st["launches"], 1))
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    traffic, traffic_source = counter_traffic()
    # the library's own rule for the two-triples-per-wave walk (bfh_bpr_set_mode "im_dual"): vdim <= 128 and >= 1024 users per queue
    dual_walk = hog == "3" and knobs.get("im_dual", "-1") != "0" and (knobs.get("im_dual", "-1") == "1" or n_local_users >= 8 * 1024)
    strict_b = 16 * D + 20 + 8 * D / (nnz / U)
    out = {
        "metric": "BPRMF training throughput (interactions/s), ML-20M-shaped synthetic, d=128",
        "value": updates / elapsed, "unit": "updates/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BPRMF sgd lr %g -> min_lr %g over %d epochs, %d negative/positive, %s-shaped synthetic (%d x %d, %d nnz%s), d=%d, "
                               "uniform sampling + verify_neg, reg 0.025, CSR + factors resident in HBM"
                               % (opt["lr"], opt["min_lr"], steps + warmup, opt["num_negative_samples"], args.shape, U, I, nnz,
                                  "" if args.scaling == "strong" or world == 1 else " per GPU", D),
                   "lr": opt["lr"], "min_lr": opt["min_lr"], "num_negative_samples": opt["num_negative_samples"], "optimizer": "sgd",
                   "parallelism": "1 GPU" if world == 1 else "dp%d: users sharded, Q replicated, %.1f delta all-reduce/epoch (inside the library, transport %s)"
                                  % (world, st["exchanges"] / max(steps, 1), ctx.transport()),
                   "hogwild": {"0": "write-through (sc1) racy stores on item rows", "1": "fp32 atomics on item rows",
                               "2": "per-XCD item-factor replicas (plain stores through the XCD's L2, merged by the delta rule "
                                    "%.1f times per epoch); popular rows stay chip-wide on fp32 atomics"
                                    % (st["merges"] / max(steps, 1)),
                               "3": "item-major walk%s: users owned by XCDs (plain stores through the owner's L2), the positive "
                                    "item row in registers with bounded-staleness atomic flushes, negatives in per-XCD replicas "
                                    "merged by the delta rule %.1f times per epoch"
                                    % (", two triples per wave" if dual_walk else "", st["merges"] / max(steps, 1))}[hog]},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": ("bpr_item_major_dual_kernel" if dual_walk else "bpr_item_major_kernel") if hog == "3" else "bpr_update_kernel",
                     "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     # SURVEY.md section 8(d)'s stricter variant, reported alongside: one side's row is read and written once per
                     # run of its triples instead of once per triple (16 d + 20 + 8 d / mean degree bytes per update)
                     "strict": {"bytes_per_update": strict_b, "achieved": achieved * strict_b / bytes_per_update,
                                "frac": achieved * strict_b / bytes_per_update / HBM_PEAK_GBS},
                     "launches_per_step": st["launches"] / max(steps, 1),
                     # the same bytes over ALL device time of a step (update launches + replica broadcast / merge kernels)
                     "frac_incl_merge_kernels": (bytes_per_update * st["samples"] / max((st["kernel_ms"] + st["aux_ms"]) * 1e-3, 1e-12)
                                                 / 1e9 / HBM_PEAK_GBS)},
        "epoch_ms": elapsed / steps * 1e3,
    }
    out.update(ctx.comm_keys())
    if traffic:
        out["roofline"]["traffic_frac"] = traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    if breakdown is not None:
        out["breakdown"] = breakdown
    if world == 1:
        try:   # the same box's HBM under a trivial kernel, next to the 8 TB/s the fraction is quoted against
            m = measured_stream_bandwidth()
            m["frac_of_triad"] = achieved / m["triad_GBps"] if m["triad_GBps"] > 0 else None
            out["roofline"]["measured_stream"] = m
            out["roofline"]["triad_GBps"], out["roofline"]["frac_of_triad"] = m["triad_GBps"], m["frac_of_triad"]
        except Exception as e:
            out["roofline"]["measured_stream"] = {"error": "%s: %s" % (type(e).__name__, e)}
    del obj
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(csr, all_cores=args.cpu_all_cores)
    if world == 1 and not args.no_extra:
        out["extra"] = run_extras(args, csr, out)
    return out


def run_extras(args, csr, out):
    """N = 1, after the timed region, in the same process: the other inner loops north_star names (ALS configs[2], WARP at configs[4]'s
    size and d), the BPRMF walk at the reference benchmark's learning rate, and the section-8(f) neighbours.  Their headline scalars
    are copied into `roofline` / `cpu_baseline` as flat keys -- the part of the line the driver's record keeps."""
    cpu = not args.no_cpu_baseline
    extra = {}
    for name, fn in (("bpr_lr005", lambda c, seed, cpu: extra_bpr_lr005(c, seed)),
                     ("als_ml20m_d128", extra_als), ("warp_ml20m_d256", extra_warp),
                     ("warp_c5_one_gpu", lambda _csr, seed, cpu: extra_warp_c5(seed, cpu=cpu)), ("topk_ml20m_d128_k100", extra_topk),
                     ("sppmi_ml20m_stream_w5", extra_sppmi), ("coo_to_csr_ml20m", extra_ingest), ("text_to_csr_2m_lines", extra_text_ingest),
                     # the top of the reference's own D-sweep (benchmark/README.md:97): d = 160, block 32 -> the wide ALS kernel (T = 5)
                     ("als_ml20m_d160", lambda c, seed, cpu: extra_als_wide(c, seed, 160))):
        if (args.only_extra and name not in args.only_extra) or name in args.skip_extra:
            continue
        try:
            extra[name] = fn(csr, args.seed, cpu=cpu)
        except Exception as e:   # the headline line is never lost to a secondary measurement
            extra[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    rf = out["roofline"]
    b = extra.get("bpr_lr005") or {}
    if "kernel_ms_per_launch" in b:
        rf.update({"bpr_lr005_kernel_ms": b["kernel_ms_per_launch"], "bpr_lr005_frac": b["frac"], "bpr_lr005_kernel_ms_max": b["kernel_ms_per_launch_max"],
                   "bpr_lr005_updates_per_s": b["updates_per_s_after_first"],
                   # |Qb| after the schedule against the reference path's: 183.0 at 8 / 64 / 128 / 256 workers alike (profiles/r06_bpr_lr005_width_and_knobs.txt);
                   # the bias norm does not depend on the schedule's width, so it is the one norm of this case a constant can stand for (seed 7, this matrix)
                   "bpr_lr005_norm_gap_Qb": b["norm_Qb"] / REFBENCH_QB_REFERENCE - 1.0})
    a = extra.get("als_ml20m_d128") or {}
    if "epoch_ms" in a:
        rf.update({"als_epoch_ms": a["epoch_ms"], "als_kernel_ms": a["kernel_ms_per_epoch"], "als_hbm_frac": a["hbm"]["frac"],
                   "als_hbm_frac_of_epoch": a["hbm"]["algorithmic_bytes_per_epoch"] / (a["epoch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "als_issued_mfma_frac_f16": a["mfma"]["frac"], "als_useful_mfma_frac": a["mfma"]["useful_frac_of_fp32_peak"],
                   "als_arith": "split-f16"})
        for k, v in (a.get("parity") or {}).items():
            rf["als_" + k] = v
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
        sr = c5.get("searching_regime") or {}
        if "epoch_ms" in sr:            # the same size with the margin lowered until the trial loop has to search (T >= 3)
            rf.update({"warp_c5_search_epoch_ms": sr["epoch_ms"], "warp_c5_search_T": sr["mean_scored_negatives_T"], "warp_c5_search_accepted_frac": sr["accepted_frac"]})
        if "agrees_with_device" in c5:   # the results check at configs[4] size (warp_results_check: oracle on the first users vs the device's rows and counts)
            rf["warp_c5_agrees_with_device"] = c5["agrees_with_device"]
        tr = c5.get("counter_traffic") or {}
        if isinstance(tr, dict) and tr.get("hbm_bytes_per_epoch"):
            dev_ms = last["trial_kernel_ms"] + last["sort_and_gather_ms"] + last["optimizer_ms"]
            rf["warp_c5_traffic_frac"] = tr["hbm_bytes_per_epoch"] / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            rf["warp_c5_traffic_T"] = tr.get("mean_scored_negatives_T")   # the T of the profiled epoch (compare with warp_c5_T)
            rf["warp_c5_traffic_source"] = "profiles/warp_pmc_latest.json (counter bytes of an earlier run) over this run's device time"
    cb = out.get("cpu_baseline") or {}
    for name, key in (("als_ml20m_d128", "als"), ("warp_ml20m_d256", "warp_ml20m"), ("warp_c5_one_gpu", "warp_c5")):
        v = (extra.get(name) or {}).get("cpu_baseline") or {}
        if "value" in v and cb:
            cb["%s_value" % key], cb["%s_unit" % key] = v["value"], v["unit"]
    return extra


def run_warp_c5(args, ctx):
    """BASELINE configs[4]: WARP on 10 M x 1 M / 1 B nnz, d = 256, USER-SHARDED over the ranks (contiguous user ranges -- every user has
    100 positives, so equal row counts are nnz-balanced), Q | Qb replicated, ONE grouped all-reduce of the item-side gradients per
    update_parameters inside the library (csrc/sgd_base.hip; warp.cc:103-201 has no multi-device form).  A step = one epoch."""
    from buffalo_amd.backend import CyWARP
    world, rank = ctx.world, ctx.rank
    U5, I5, deg, d = 10_000_000, 1_000_000, 100, WARP_D
    if args.c5_users:
        U5 = args.c5_users
    u0, u1 = U5 * rank // world, U5 * (rank + 1) // world
    indptr, keys, P, Q, Qb = warp_c5_inputs(u0=u0, u1=u1, users=U5)
    total_nnz = U5 * deg
    g = CyWARP()
    g.set_device(ctx.local_rank)
    path = _opt_file(dict(WARP_OPT, num_iters=args.steps + args.warmup))
    assert g.init(path)
    os.unlink(path)
    g.sync_every_epoch = False
    g.initialize_model(P, Q, Qb, total_nnz, True)
    g.set_resident_csr(indptr, keys)
    g.set_shard(u0 * deg, world)
    comm = None
    if world > 1:
        comm = ctx.make_comm()
        g.set_comm(comm)
    n_local = u1 - u0
    # results check at the size that runs (one GPU, rank 0): the accumulated gradient rows of the first users after the FIRST add_jobs of this object, read
    # from HBM (a 20 MB copy once, in the first step -- a warm-up step unless --warmup 0), against the oracle afterwards (warp_results_check)
    n_chk = min(n_local, int(os.environ.get("BFH_WARP_CHECK_USERS", "20000")))
    chk = {}
    want_check = world == 1 and not args.no_cpu_baseline
    P0 = np.ascontiguousarray(P[:n_chk]).copy() if want_check else None
    Q0 = Q.copy() if want_check else None

    def step():
        g.add_jobs(0, n_local, indptr, None)
        if want_check and "G" not in chk:
            chk["G"] = g.device_tensor("gradP", (n_local, d))[:n_chk].cpu().numpy().copy()
        g.update_parameters()

    elapsed = timed_steps(ctx, step, args.steps, args.warmup, before_timing=g.reset_stats)
    st = g.stats()
    lo, hi = ctx.minmax([st["kernel_ms"], st["aux_ms"], st["optimizer_ms"], st.get("allreduce_ms", 0.0), float(st["scored_negatives"]), float(st["accepted"])])
    tot_scored = st["scored_negatives"]
    if ctx.dist is not None:
        t = ctx.torch.tensor([float(st["scored_negatives"]), float(st["accepted"]), float(st.get("loaded_rows", 0))], dtype=ctx.torch.float64)
        ctx.dist.all_reduce(t)
        tot_scored, tot_acc, tot_loaded = (float(x) for x in t)
    else:
        tot_acc, tot_loaded = float(st["accepted"]), float(st.get("loaded_rows", 0))
    del g
    if rank != 0:
        return None
    K = max(args.steps, 1)
    row = 4 * d
    positives = float(total_nnz) * args.steps
    alg = (8 * tot_acc + 2 * (positives - tot_acc) + tot_scored) * row + 4 * positives        # SURVEY 8(d), all steps, all ranks
    dev_ms = (hi[0] + hi[1]) / K                                                             # slowest rank's trial + sort/gather kernels per step
    tr = _warp_counter_traffic("c5") or {}
    out = {"metric": "WARP training throughput (positives/s), configs[4]-shaped synthetic (10M x 1M, 1B nnz), d=256",
           "value": positives / elapsed, "unit": "positives/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "WARP adagrad lr %g, dot score, max_trials %d, threshold %g, %d x %d, %d nnz (100 per user, popularity-skewed bands), d=%d, "
                                  "everything resident in HBM" % (WARP_OPT["lr"], WARP_OPT["max_trials"], WARP_OPT["threshold"], U5, I5, total_nnz, d),
                      "lr": WARP_OPT["lr"], "optimizer": "adagrad",
                      "parallelism": "1 GPU" if world == 1 else "dp%d: users sharded (%d per rank), Q | Qb replicated, one grouped gradient all-reduce per "
                                     "epoch inside the library, transport %s" % (world, n_local, ctx.transport())},
           "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                        # SURVEY 8(d)'s formula counts six gradient-row read-modify-writes per positive that the gather-by-item formulation never
                        # performs, so this figure can exceed the peak: it is the contract's number, NOT evidence of bandwidth -- see traffic_frac
                        "achieved": alg / K / world / (dev_ms * 1e-3) / 1e9, "frac": alg / K / world / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "achieved_is": "SURVEY 8(d) algorithmic bytes per rank and step over the slowest rank's trial + sort + gather kernel time",
                        "kernel": "warp_update_kernel + grad_gather_kernel", "kernel_ms": dev_ms, "trial_kernel_ms": hi[0] / K, "sort_and_gather_ms": hi[1] / K,
                        "optimizer_ms": hi[2] / K, "allreduce_ms": hi[3] / K, "mean_scored_negatives_T": tot_scored / positives,
                        "accepted_frac": tot_acc / positives, "candidate_rows_fetched_per_positive": tot_loaded / positives,
                        "traffic": (tr.get("hbm_bytes_per_epoch") if world == 1 else None),
                        "traffic_source": "profiles/warp_pmc_latest.json (rocprofv3 --pmc passes of an earlier one-GPU run of this workload)"},
           "breakdown": {"trial_kernel_ms_min_rank": lo[0] / K, "trial_kernel_ms": hi[0] / K, "sort_and_gather_ms": hi[1] / K, "optimizer_ms": hi[2] / K,
                         "allreduce_ms": hi[3] / K, "what": "per step; max over ranks unless named min"}}
    out.update(ctx.comm_keys())
    if world == 1 and tr.get("hbm_bytes_per_epoch"):
        out["roofline"]["traffic_frac"] = tr["hbm_bytes_per_epoch"] / ((hi[0] + hi[1] + hi[2]) / K * 1e-3) / 1e9 / HBM_PEAK_GBS
    if world == 1 and not args.no_cpu_baseline:
        n100 = max(1, (u1 - u0) // 100)
        out["cpu_baseline"] = _warp_cpu_baseline(indptr[:n100], keys[:int(indptr[n100 - 1])], I5, d, args.seed, seconds=10.0)
        out["cpu_baseline"]["sample"] += "; the first 1/100 of the users against the full item table"
        try:
            rc = warp_results_check(indptr, keys, P0, Q0, np.zeros((I5, 1), np.float32), chk["G"], n_chk, total_nnz, dict(WARP_OPT, num_iters=args.steps + args.warmup))
            out["results_check"] = rc
            out["roofline"]["agrees_with_device"] = rc["agrees_with_device"]
        except Exception as e:  # noqa: BLE001 -- reported, never silently dropped
            out["results_check"] = {"error": "%s: %s" % (type(e).__name__, e)}
            out["roofline"]["agrees_with_device"] = False
    return out


def run_als(args, ctx):
    """BASELINE configs[2] over N ranks: ALS (iALS++ at d = 128) on the ML-20M shape; the rows of a half-epoch sharded nnz-balanced, both factor
    matrices and both CSR orientations replicated, the solved row blocks published after each half-epoch (bfh_als_publish_rows).  A step = one epoch."""
    from buffalo_amd import ingest, synth
    from buffalo_amd.backend import CyALS
    from buffalo_amd.dist import CommDataParallelALS
    world, rank = ctx.world, ctx.rank
    csr = load_matrix(args.shape, args.seed)
    U, I, nnz = csr.num_users, csr.num_items, csr.nnz
    vals = (1 + np.random.default_rng(args.seed).poisson(1.0, size=nnz)).astype(np.float32)
    col = ingest.coo_to_csr(csr.keys, csr.rows(), vals, I, U)
    P, Q, _ = synth.init_factors(U, I, D, seed=args.seed)
    g = CyALS()
    g.set_device(ctx.local_rank)
    path = _opt_file(ALS_OPT)
    assert g.init(path)
    os.unlink(path)
    g.initialize_model(P, Q)
    g.set_resident_csr(0, csr.indptr, csr.keys, vals)
    g.set_resident_csr(1, col["indptr"], col["key"], col["val"])
    g.set_mode("als_writeback", 0)
    if world > 1:
        comm = ctx.make_comm()
        g.set_comm(comm)
        dp = CommDataParallelALS(g, comm, (csr.indptr, col["indptr"]), U, I)
        step = dp.epoch
    else:
        def step():
            g.precompute(0)
            g.partial_update(0, U, csr.indptr, None, None, 0)
            g.precompute(1)
            g.partial_update(0, I, col["indptr"], None, None, 1)
    elapsed = timed_steps(ctx, step, args.steps, args.warmup, before_timing=g.reset_stats)
    st = g.stats()
    lo, hi = ctx.minmax([st["kernel_ms"], st["aux_ms"]])
    del g
    if rank != 0:
        return None
    K = max(args.steps, 1)
    alg_bytes = 2 * nnz * (4 * D + 8) + (U + I) * (8 * D + 8) + (U + I) * 4 * D     # SURVEY 8(d) B_als, both half-epochs, whole job
    kernel_s = hi[0] / K * 1e-3
    return {**ctx.comm_keys(),
            "metric": "ALS training throughput (interactions/s), ML-20M-shaped synthetic, d=128 (iALS++)",
            "value": 2.0 * nnz * args.steps / elapsed, "unit": "interactions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32 (split-f16 Gramian)",
            "data": "synthetic",
            "config": {"workload": "ALS iALS++ (block 32, 3 CG steps, alpha 8, reg 0.1), %s-shaped synthetic (%d x %d, %d nnz, values 1+Poisson(1)), d=%d, "
                                   "Gramian: split-f16 (~22-bit products, fp32 accumulate)" % (args.shape, U, I, nnz, D),
                       "parallelism": "1 GPU" if world == 1 else "dp%d: rows of each half-epoch sharded nnz-balanced, factors replicated, solved rows "
                                      "published per half-epoch inside the library, transport %s" % (world, ctx.transport())},
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": alg_bytes / world / kernel_s / 1e9,
                         "frac": alg_bytes / world / kernel_s / 1e9 / HBM_PEAK_GBS, "kernel": "als_pc_kernel", "kernel_ms": hi[0] / K,
                         "kernel_ms_min_rank": lo[0] / K, "aux_ms": hi[1] / K, "traffic": None,
                         "achieved_is": "SURVEY 8(d) B_als per rank over the slowest rank's row-kernel time per epoch"}}


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: N ranks of this file (one per GPU), rendezvous on 127.0.0.1; rank 0 prints the line.
    (The driver's `python -m torch.distributed.run ... bench.py --gpus N` sets WORLD_SIZE and never gets here.)"""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0 and rc == 0:
                rc = r
                for q in live:              # a rank died: the others would wait in a collective forever (exact PIDs, never a pattern)
                    q.terminate()
    return rc


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 250 (bpr: ~2.1 s timed region on one MI355X), 20 (als), 5 (warp_c5)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["bpr", "warp_c5", "als"], default="bpr",
                    help="bpr = the headline (configs[1] / configs[3]); warp_c5 = configs[4], user-sharded; als = configs[2], rows sharded")
    ap.add_argument("--shape", default="ml20m")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--lr", type=float, default=None, help="BPRMF learning rate (default: the option default 0.002)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--minibatches", type=int, default=1, help="all-reduce points per epoch (N>1)")
    ap.add_argument("--c5-users", type=int, default=0, help="warp_c5: number of users instead of 10 M (tests on shared boxes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-all-cores", action="store_true", help="also time the oracle with one worker per host core (6 s on the 256-core box; off by default)")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary measurements (N=1)")
    ap.add_argument("--only-extra", action="append", default=[], help="run only these extras (name as in bench_extra.json's `extra`)")
    ap.add_argument("--skip-extra", action="append", default=[], help="leave these extras out (scripts/gpu_profile.sh: the BPRMF lr-0.05 extra launches the "
                                                                     "headline kernel under other settings, which would mix into its per-kernel average)")
    ap.add_argument("--mode", action="append", default=[], help="backend knob name=value (e.g. hogwild_atomic=0)")
    return ap.parse_args(argv)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if os.environ.get("BFH_COMM_TRANSPORT") == "shm":      # one-GPU rehearsal of --gpus N: the shared-memory TEST transport lives in
        os.environ.setdefault("BFH_LIBRARY", "test")       # libbuffalo_hip_test.so only (buffalo_amd/_lib.py); the ranks inherit the choice
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))
    if args.steps is None:
        args.steps = {"bpr": 250, "als": 20, "warp_c5": 5}[args.workload]
    ctx = Ctx(args)
    out = {"bpr": run_bpr, "warp_c5": run_warp_c5, "als": run_als}[args.workload](args, ctx)
    if ctx.rank == 0:
        emit(out)
    ctx.close()


if __name__ == "__main__":
    main()
