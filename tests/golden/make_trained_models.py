"""Whole training runs of STOCK buffalo's Python over the oracle, as golden fixtures (TEST INFRASTRUCTURE).

    python tests/golden/make_trained_models.py [--out FILE]      # needs /root/reference; default tests/golden/trained_models.npz

The reference's `ALS` / `EALS` fronts (CPU mode), its MatrixMarket loader (over its compiled fileio.hpp and the in-memory h5py:
make_data_vectors.install) and its evaluation run UNMODIFIED from /root/reference; the compiled training classes are the oracle's
(`OracleALS` / `OracleEALS` where `buffalo.algo._als.CyALS` / `_eals.CyEALS` are imported).  From a seeded coordinate file and
seeded np.random the whole flow is deterministic: validation draw, |N(0, 1/d^2)| factors, epochs, metrics.  Stored per case: the
factors after training, what `train()` returned, what `get_validation_results()` says afterwards.

tests/test_trained_models_ref.py rebuilds the same inputs and trains the stand-in fronts (tests/front_harness) from the same seeds:
over the oracle on CPU -- where every number must be IDENTICAL, front for front -- and over the HIP backend on a GPU box, where
the factors must agree within the parity tolerance of the kernels.  d is a multiple of 32 in the ALS cases so that the
accelerator path's padded width (ceil32) draws the same random numbers as the CPU path's (als.py:81).
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
OUT = os.path.join(HERE, "trained_models.npz")

CASES = {
    # name: (algo, (U, I, seed of the file), options, np.random seed before the data is built)
    "als_llt_d32": ("als", (150, 90, 1), dict(d=32, num_iters=3, optimizer="llt", random_seed=7, num_workers=2, validation={"topk": 10}), 5),
    "als_manual_cg_d64": ("als", (150, 90, 2), dict(d=64, num_iters=4, optimizer="manual_cg", random_seed=9, num_workers=2, alpha=4.0,
                                                     reg_u=0.05, reg_i=0.2, validation={"topk": 10}), 6),
    # a WELL-POSED whole run on the reference's default solver family: d = 64 over 2,000 items, twelve CG steps, regulariser 5 --
    # the oracle stays within 1e-4 of the float64 recurrence through all four epochs, so a device run must reach this model to 5e-3.
    # (The three-step case above cannot be held that way: from the |N(0, 1/d^2)| start its first user half-epoch returns rows of
    # size ~y / reg, the next system has a condition number beyond fp32, and the oracle ITSELF ends 2 % from the float64 recurrence.)
    # One worker: with two, the per-thread loss sums of 3,200 rows are combined in scheduling order and the reported loss wobbles in
    # its last digit from run to run (the factors do not).
    "als_manual_cg_d64_wellposed": ("als", (1200, 2000, 12, 0.10, 0.01), dict(d=64, num_iters=4, optimizer="manual_cg", num_cg_max_iters=12,
                                                                             random_seed=9, num_workers=1, alpha=1.0, reg_u=5.0, reg_i=5.0,
                                                                             validation={"topk": 10}), 11),
    "eals_d16": ("eals", (150, 90, 3), dict(d=16, num_iters=4, random_seed=3, num_workers=2, c0=64.0, exponent=0.5, validation={"topk": 10}), 7),
    # the SGD fronts in ACCELERATOR mode (the path this repository replaces): `CuBPRMF` / the WARP scaffold's object is the oracle
    # behind the accelerator's method surface, in its deterministic modes (counter sampler, CSR order, jobs processed inside
    # add_jobs with the lr of completed work) -- the modes the HIP backend's `sequential` / frozen-epoch paths reproduce to 1e-5 / 1e-4
    "bpr_sgd_d20": ("bpr", (150, 90, 4), dict(d=20, num_iters=4, lr=0.05, min_lr=0.002, random_seed=7, num_workers=1, accelerator=True,
                                               evaluation_period=2, validation={"topk": 10}), 8),
    "bpr_adagrad_d40": ("bpr", (150, 90, 5), dict(d=40, num_iters=3, lr=0.05, optimizer="adagrad", random_seed=11, num_workers=1,
                                                   accelerator=True, evaluation_period=3, validation={"topk": 10}), 9),
    "warp_d24": ("warp", (150, 90, 6), dict(d=24, num_iters=3, lr=0.05, max_trials=20, threshold=0.5, random_seed=13, num_workers=1,
                                             evaluation_period=3, validation={"topk": 10}), 10),
}
DETERMINISTIC = dict(sampler="counter", pos_order="csr", inline=True)


def accelerator_over_oracle(oracle_cls):
    """The accelerator classes' method surface (cuda/_bpr.pyx:37-80) on an oracle class in its deterministic modes: no padding
    (`get_vdim` = d), nothing to announce, the model is trained in place."""
    class OracleBehindTheAcceleratorSurface(oracle_cls):
        def init(self, opt_path):
            path = opt_path.decode("utf-8") if isinstance(opt_path, bytes) else opt_path
            with open(path) as f:
                self._d = json.load(f)["d"]
            return super().init(path)

        def get_vdim(self):
            return self._d

        def set_placeholder(self, *args):
            pass

        def initialize_model(self, P, Q, Qb, num_total_samples, set_gpu=False):
            super().initialize_model(P, Q, Qb, num_total_samples)
            if set_gpu:                       # the hand-over right before training (bpr.py:207)
                self.set_modes(**DETERMINISTIC)
                self.launch_workers()         # inline mode: seeds the stream, starts no thread

        def synchronize(self, *args):
            pass
    return OracleBehindTheAcceleratorSurface


def coordinate_text(U, I, seed, p_in=0.45, p_out=0.04):
    """A planted coordinate file: 6 taste groups, in-group cells likelier, integer values 1..5, lines in no particular order."""
    rng = np.random.default_rng(seed)
    ug, ig = rng.integers(0, 6, U), rng.integers(0, 6, I)
    p = np.where(ug[:, None] == ig[None, :], p_in, p_out)
    rows, cols = np.nonzero(rng.random((U, I)) < p)
    order = rng.permutation(len(rows))
    rows, cols = rows[order], cols[order]
    vals = rng.integers(1, 6, len(rows))
    return ("%%MatrixMarket matrix coordinate integer general\n%\n" + "%d %d %d\n" % (U, I, len(rows))
            + "".join("%d %d %d\n" % (r + 1, c + 1, v) for r, c, v in zip(rows, cols, vals)))


def data_option(opt_cls, path, work=None):
    opt = opt_cls().get_default_option()
    opt.input.main = path
    opt.input.uid = None
    opt.input.iid = None
    opt.data.validation = {"name": "sample", "p": 0.05, "max_samples": 60}
    if work is not None:
        opt.data.tmp_dir = work
        opt.data.path = os.path.join(work, "db.h5py")
    return opt


def reference_models():
    import make_data_vectors as M
    from oracle import oracle
    M.install()
    sys.modules["buffalo.algo._als"].CyALS = oracle.OracleALS
    sys.modules["buffalo.algo._eals"].CyEALS = oracle.OracleEALS
    sys.modules["buffalo.algo.cuda._bpr"].CyBPR = accelerator_over_oracle(oracle.OracleBPRMF)     # bpr.py:19 imports it as CuBPRMF
    sys.modules["buffalo.algo._warp"].CyWARP = accelerator_over_oracle(oracle.OracleWARP)          # warp.py has no accelerator class: see below
    sys.modules["buffalo.parallel._core"].dot_topn = oracle.dot_topn
    sys.modules["buffalo.parallel._core"].quickselect = oracle.quickselect
    from buffalo.algo.als import ALS
    from buffalo.algo.bpr import BPRMF
    from buffalo.algo.eals import EALS
    from buffalo.algo.options import ALSOption, BPRMFOption, EALSOption, WARPOption
    from buffalo.algo.warp import WARP
    fronts = {"als": (ALS, ALSOption), "eals": (EALS, EALSOption), "bpr": (BPRMF, BPRMFOption), "warp": (WARP, WARPOption)}
    from buffalo.data.mm import MatrixMarketOptions
    from buffalo.misc import aux, log
    log.set_log_level(1)
    out, meta = {}, {}
    for name, (algo, shape, over, np_seed) in CASES.items():
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "main.mtx")
            with open(path, "w") as f:
                f.write(coordinate_text(*shape))
            cls, opt_cls = fronts[algo]
            opt = opt_cls().get_default_option()
            opt.update(over)
            opt.validation = aux.Option(over["validation"])
            np.random.seed(np_seed)
            model = cls(opt, data_opt=aux.Option(data_option(MatrixMarketOptions, path, d)))
            model.initialize()
            if algo == "warp":
                # warp.py:31-32 refuses accelerator = True in the constructor but carries the accelerator scaffold in _prepare_train /
                # _finalize_train (:212-234): built without it, switched before train() (as in make_front_traces.py)
                model.opt.accelerator = True
            ret = model.train()
            vali = model.get_validation_results()
            out[name + "/P"], out[name + "/Q"] = np.array(model.P), np.array(model.Q)
            if hasattr(model, "Qb"):
                out[name + "/Qb"] = np.array(model.Qb)
            meta[name] = {"train": {k: float(v) for k, v in ret.items()}, "validation": {k: float(v) for k, v in vali.items()},
                          "header": {k: int(v) for k, v in model.data.get_header().items()}}
    out["meta"] = np.array(json.dumps(meta, sort_keys=True))
    return out


if __name__ == "__main__":
    path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else OUT
    vec = reference_models()
    np.savez_compressed(path, **vec)
    meta = json.loads(str(vec["meta"]))
    for k, v in meta.items():
        print(k, v["train"], {a: round(b, 4) for a, b in v["validation"].items()})
    print("wrote", path, os.path.getsize(path), "bytes")
