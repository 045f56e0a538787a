"""Golden call traces of the drop-in boundary, made with the REFERENCE's own Python fronts.

`buffalo/algo/bpr.py` and `buffalo/algo/als.py` are pure Python above their Cython classes.  Here they are imported unmodified
from /root/reference (a namespace stub keeps `buffalo/__init__.py` from running; the compiled extensions they import --
`_log`, `_bpr`, `_als`, ..., `parallel._core`, `data.fileio`, `h5py` -- are replaced by empty stubs), given an in-memory matrix
through the reference's real `Data` / `BufferedDataMatrix` classes (the HDF5 handle is a dict look-alike), and run with
`accelerator = True` against a backend that only RECORDS what it is asked to do: the methods of `CuBPR` / `CuALS`
(/root/reference/buffalo/algo/cuda/_bpr.pyx:27-80, _als.pyx:25-67) with their scalar arguments and a digest of every array.

The result is what a replacement for those Cython classes actually receives from stock buffalo: option file, model binding,
placeholder, chunk boundaries, loss samples, call order.  `tests/test_front_trace_cpu.py` replays the same cases through the
stand-in front of tests/front_harness (and, where /root/reference exists, regenerates the traces) and requires identical
traces.  Run from the repo root:  python tests/golden/make_front_traces.py
"""
import json
import os
import sys
import types
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
OUT = os.path.join(HERE, "front_traces.json")


# ------------------------------------------------------------------------------------------------
# the recording backend (shared with the test)
# ------------------------------------------------------------------------------------------------
def digest(a):
    a = np.asarray(a)
    return {"dtype": str(a.dtype), "shape": list(a.shape), "c_contiguous": bool(a.flags["C_CONTIGUOUS"]),
            "crc32": zlib.crc32(np.ascontiguousarray(a).tobytes())}


def summarize(x):
    if isinstance(x, np.ndarray):
        return digest(x)
    if isinstance(x, (bytes, str)):
        path = x.decode() if isinstance(x, bytes) else x
        if os.path.isfile(path):                       # init(opt_path): the option file's content, not its temporary name
            opt = json.load(open(path))
            opt.pop("data_opt", None)
            return {"option_file": opt}
        return path
    if isinstance(x, (bool, np.bool_)):
        return bool(x)
    if isinstance(x, (int, np.integer)):
        return int(x)
    if isinstance(x, (float, np.floating)):
        return float(x)
    return repr(x)


class Recorder:
    """Stands where CuBPR / CuALS stand.  Every call is appended to `trace`; return values are fixed (the fronts only look at
    init's bool, get_vdim and the losses)."""
    trace = None    # set per run

    def __init__(self, *a, **k):
        self._d = None

    def _rec(self, name, args):
        Recorder.trace.append({"call": name, "args": [summarize(a) for a in args]})

    def init(self, opt_path):
        self._rec("init", (opt_path,))
        self._d = json.load(open(opt_path.decode() if isinstance(opt_path, bytes) else opt_path))["d"]
        return True

    def get_vdim(self):
        self._rec("get_vdim", ())
        return (self._d + 31) // 32 * 32                # bpr.cu:266-267, als.cu:251-252

    def _chunk(self, name, start_x, next_x, indptr, *bufs):
        # BufferedDataMatrix hands over its fixed-size key / value buffers; a backend may read the first
        # indptr[next_x - 1] - indptr[start_x - 1] entries (buffered_data.py:112-121), what lies behind them is left over from
        # earlier chunks.  The digest covers what may be read; the buffer length is recorded next to it.
        size = int(indptr[next_x - 1] - (indptr[start_x - 1] if start_x else 0))
        return [int(start_x), int(next_x), digest(indptr)] + [dict(digest(b[:size]), buffer_len=int(b.shape[0])) for b in bufs]

    def add_jobs(self, start_x, next_x, indptr, keys):
        Recorder.trace.append({"call": "add_jobs", "args": self._chunk("add_jobs", start_x, next_x, indptr, keys)})

    def compute_loss(self, *args):
        self._rec("compute_loss", args)
        return 0.625

    def partial_update(self, start_x, next_x, indptr, keys, vals, axis):
        Recorder.trace.append({"call": "partial_update", "args": self._chunk("partial_update", start_x, next_x, indptr, keys, vals) + [int(axis)]})
        return 3.0, 4.0

    def partial_update_user(self, *args):
        self._rec("partial_update_user", args)
        return 1.5

    def partial_update_item(self, *args):
        self._rec("partial_update_item", args)
        return 2.5

    def partial_update_context(self, *args):
        self._rec("partial_update_context", args)
        return 0.75

    def update(self, *args):                             # CyEALS.update -> bool (eals.py:131)
        self._rec("update", args)
        return True

    def estimate_loss(self, *args):                      # CyEALS.estimate_loss -> (rmse, total)
        self._rec("estimate_loss", args)
        return 0.5, 12.0

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)

        def call(*args):
            self._rec(name, args)
        return call


# ------------------------------------------------------------------------------------------------
# the cases: matrix, options
# ------------------------------------------------------------------------------------------------
def case_matrix(U, I, density, seed):
    rng = np.random.default_rng(seed)
    rows, cols = np.nonzero(rng.random((U, I)) < density)
    for u in range(U):                                  # no empty rows: the fronts would skip them in different places
        if not np.any(rows == u):
            rows, cols = np.append(rows, u), np.append(cols, rng.integers(I))
    for i in range(I):
        if not np.any(cols == i):
            rows, cols = np.append(rows, rng.integers(U)), np.append(cols, i)
    order = np.lexsort((cols, rows))
    rows, cols = rows[order].astype(np.int32), cols[order].astype(np.int32)
    vals = (1 + rng.poisson(1.0, size=rows.shape[0])).astype(np.float32)
    return U, I, rows, cols, vals


def groups_of(U, I, rows, cols, vals):
    def group(major, minor, v, n):
        order = np.lexsort((minor, major))
        return {"indptr": np.cumsum(np.bincount(major, minlength=n)).astype(np.int64), "key": minor[order].astype(np.int32),
                "val": v[order].astype(np.float32)}
    return {"rowwise": group(rows, cols, vals, U), "colwise": group(cols, rows, vals, I)}


CASES = {
    # name: (algo, matrix (U, I, density, seed), batch_mb, option overrides)
    "bpr_one_chunk": ("bpr", (60, 40, 0.15, 1), 1024, dict(d=20, num_iters=2, random_seed=11, compute_loss_on_training=True)),
    "bpr_chunked_no_loss": ("bpr", (90, 50, 0.2, 2), 0.004, dict(d=32, num_iters=3, random_seed=5, compute_loss_on_training=False,
                                                                   sampling_power=1.0, use_bias=False)),
    "als_one_chunk": ("als", (60, 40, 0.15, 3), 1024, dict(d=20, num_iters=2, random_seed=7)),
    "als_chunked": ("als", (90, 50, 0.2, 4), 0.004, dict(d=40, num_iters=2, random_seed=9, compute_loss_on_training=False)),
    # WARP: the reference's front refuses accelerator = True in its constructor (warp.py:30-32, "not implemented yet") but carries the
    # same accelerator scaffold as BPRMF in _prepare_train / _finalize_train (warp.py:212-234).  The object is built with
    # accelerator = False and switched before train(): the trace is what that scaffold would ask of a GPU WARP.
    "warp_scaffold": ("warp", (60, 40, 0.15, 6), 1024, dict(d=24, num_iters=2, random_seed=13, compute_loss_on_training=True)),
    "warp_scaffold_chunked": ("warp", (90, 50, 0.2, 7), 0.004, dict(d=64, num_iters=2, random_seed=14, compute_loss_on_training=False)),
}


# ------------------------------------------------------------------------------------------------
# the reference's fronts
# ------------------------------------------------------------------------------------------------
def install_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
    pkg = types.ModuleType("buffalo")
    pkg.__path__ = [os.path.join(REF, "buffalo")]
    sys.modules["buffalo"] = pkg

    class PyBuffaloLog:
        def set_log_level(self, lvl):
            pass

        def get_log_level(self):
            return 1
    stub("buffalo.misc._log", PyBuffaloLog=PyBuffaloLog)
    stub("h5py", File=object)
    stub("buffalo.data.fileio", chunking_into_bins=None, parallel_build_sppmi=None, sort_and_compressed_binarization=None)
    for mod, cls in (("buffalo.algo._bpr", "CyBPRMF"), ("buffalo.algo._als", "CyALS"), ("buffalo.algo._warp", "CyWARP"),
                     ("buffalo.algo._cfr", "CyCFR"), ("buffalo.algo._eals", "CyEALS"), ("buffalo.algo._plsi", "CyPLSI"),
                     ("buffalo.algo._w2v", "CyW2V")):
        stub(mod, **{cls: Recorder})
    stub("buffalo.algo.cuda", __path__=[])
    stub("buffalo.algo.cuda._bpr", CyBPR=Recorder)      # bpr.py:19: from buffalo.algo.cuda._bpr import CyBPR as CuBPRMF
    stub("buffalo.algo.cuda._als", CyALS=Recorder)
    stub("buffalo.parallel._core", quickselect=None, dot_topn=None)


class FakeH5(dict):
    """What the reference's Data methods touch of an h5py.File: groups by name, `attrs`."""
    def __init__(self, groups, attrs):
        super().__init__(groups)
        self.attrs = attrs


def reference_trace(name):
    from buffalo.algo.als import ALS
    from buffalo.algo.bpr import BPRMF
    from buffalo.algo.options import ALSOption, BPRMFOption, WARPOption
    from buffalo.algo.warp import WARP
    from buffalo.data.base import Data
    from buffalo.data.mm import MatrixMarketOptions
    algo, shape, batch_mb, over = CASES[name]
    U, I, rows, cols, vals = case_matrix(*shape)

    class MemData(Data):
        name = "MemData"

        def create_database(self, filename, **kwargs):
            pass
    dopt = MatrixMarketOptions().get_default_option()
    dopt.data.batch_mb = batch_mb
    data = MemData(dopt)
    data.data_type = "matrix"
    data.handle = FakeH5(groups_of(U, I, rows, cols, vals), {"num_nnz": len(rows), "num_users": U, "num_items": I, "completed": 1})
    opt = {"bpr": BPRMFOption, "als": ALSOption, "warp": WARPOption}[algo]().get_default_option()
    opt.update(over)
    opt.update(dict(accelerator=algo != "warp", validation={}, evaluation_on_learning=False, save_best=False, num_workers=2))
    Recorder.trace = []
    model = {"bpr": BPRMF, "als": ALS, "warp": WARP}[algo](opt, data=data)
    model.initialize()
    if algo == "warp":
        model.opt.accelerator = True
    ret = model.train()
    shapes = {k: list(getattr(model, k).shape) for k in ("P", "Q")}
    return {"trace": Recorder.trace, "train_returned": {k: float(v) for k, v in ret.items()}, "final_shapes": shapes}


# ------------------------------------------------------------------------------------------------
# CFR and EALS fronts (cfr.py, eals.py): same recording stand-in where CyCFR / CyEALS stand
# ------------------------------------------------------------------------------------------------
MORE_CASES = {
    # name: (algo, matrix, batch_mb, option overrides)
    "cfr_one_range": ("cfr", (50, 30, 0.2, 5), 1024, dict(d=12, num_iters=2, random_seed=3)),
    "cfr_ranges": ("cfr", (70, 40, 0.25, 6), 0.006, dict(d=16, num_iters=2, random_seed=4, l=0.5)),
    "eals": ("eals", (50, 30, 0.2, 7), 1024, dict(d=12, num_iters=3, random_seed=2, c0=2.0, exponent=0.5)),
}


def sppmi_like(I, seed):
    """A symmetric item x item matrix with positive values in the layout of the `sppmi` group (what the matrix holds does not
    matter to a call trace)."""
    rng = np.random.default_rng(seed)
    a, b = np.nonzero(np.triu(rng.random((I, I)) < 0.2, 1))
    rows, cols = np.concatenate([a, b]), np.concatenate([b, a])
    v = rng.random(len(a)).astype(np.float32) + 0.1
    vals = np.concatenate([v, v])
    order = np.lexsort((cols, rows))
    return {"indptr": np.cumsum(np.bincount(rows, minlength=I)).astype(np.int64), "key": cols[order].astype(np.int32),
            "val": vals[order].astype(np.float32)}


def reference_trace_more(name):
    from buffalo.algo.cfr import CFR
    from buffalo.algo.eals import EALS
    from buffalo.algo.options import CFROption, EALSOption
    from buffalo.data.base import Data
    from buffalo.data.mm import MatrixMarketOptions
    from buffalo.data.stream import StreamOptions
    algo, shape, batch_mb, over = MORE_CASES[name]
    U, I, rows, cols, vals = case_matrix(*shape)

    class MemData(Data):
        name = "MemData"

        def create_database(self, filename, **kwargs):
            pass
    dopt = (StreamOptions if algo == "cfr" else MatrixMarketOptions)().get_default_option()
    dopt.data.batch_mb = batch_mb
    if algo == "cfr":
        dopt.data.internal_data_type = "matrix"           # cfr.py:52: the stream loader in matrix layout
    data = MemData(dopt)
    data.data_type = "stream" if algo == "cfr" else "matrix"
    groups = groups_of(U, I, rows, cols, vals)
    attrs = {"num_nnz": len(rows), "num_users": U, "num_items": I, "completed": 1}
    if algo == "cfr":
        groups["sppmi"] = sppmi_like(I, shape[3])
        attrs["sppmi_nnz"] = len(groups["sppmi"]["key"])
    data.handle = FakeH5(groups, attrs)
    opt = (CFROption if algo == "cfr" else EALSOption)().get_default_option()
    opt.update(over)
    opt.update(dict(validation={}, evaluation_on_learning=False, save_best=False))
    Recorder.trace = []
    model = (CFR if algo == "cfr" else EALS)(opt, data=data)
    model.initialize()
    ret = model.train()
    return {"trace": Recorder.trace, "train_returned": {k: float(v) for k, v in ret.items()}}


# ------------------------------------------------------------------------------------------------
# validation metrics (evaluate/base.py:44-148) by the reference's own code
# ------------------------------------------------------------------------------------------------
def np_quickselect(scores, result, sorted, num_threads):
    """Stand-in for parallel::quickselect on distinct scores: the k best columns of every row, best first."""
    result[:] = np.argsort(-scores, axis=1, kind="stable")[:, :result.shape[1]]


def metrics_case():
    """60 x 40 matrix; one held-out entry for two users in three (everybody keeps other entries); seeded N(0, 1) factors."""
    U, I, rows, cols, vals = case_matrix(60, 40, 0.2, 8)
    rng = np.random.default_rng(21)
    held = np.zeros(len(rows), dtype=bool)
    for u in range(U):
        idx = np.flatnonzero(rows == u)
        if u % 3 and len(idx) > 1:
            held[rng.choice(idx)] = True
    vali = {"row": rows[held].astype(np.int32), "col": cols[held].astype(np.int32), "val": vals[held].astype(np.float32)}
    P = rng.normal(size=(U, 20)).astype(np.float32)
    Q = rng.normal(size=(I, 20)).astype(np.float32)
    return U, I, rows[~held], cols[~held], vals[~held], vali, P, Q


class FakeGroup(dict):
    def __init__(self, d, attrs):
        super().__init__(d)
        self.attrs = attrs


def reference_metrics():
    import buffalo.evaluate.base as ev
    from buffalo.algo.als import ALS
    from buffalo.algo.options import ALSOption
    from buffalo.data.base import Data
    from buffalo.data.mm import MatrixMarketOptions
    from buffalo.misc import aux
    ev.quickselect = np_quickselect
    U, I, rows, cols, vals, vali, P, Q = metrics_case()

    class MemData(Data):
        name = "MemData"

        def create_database(self, filename, **kwargs):
            pass
    data = MemData(MatrixMarketOptions().get_default_option())
    data.data_type = "matrix"
    groups = groups_of(U, I, rows, cols, vals)
    groups["vali"] = FakeGroup(vali, {"num_samples": len(vali["row"])})
    data.handle = FakeH5(groups, {"num_nnz": len(rows), "num_users": U, "num_items": I, "completed": 1})
    opt = ALSOption().get_default_option()
    opt.update(dict(d=20, accelerator=True, num_workers=1, validation=aux.Option({"topk": 10, "batch": 16, "eval_samples": 0})))
    Recorder.trace = []
    model = ALS(opt, data=data)
    model.initialize()
    model.P, model.Q = P.copy(), Q.copy()
    out = {}
    for topk in (10, 25):
        model.opt.validation.topk = topk
        out["topk%d" % topk] = {k: float(v) for k, v in model.get_validation_results().items()}
    return out


# ------------------------------------------------------------------------------------------------
# ParALS / ParBPRMF (parallel/base.py:77-156) around a recording dot_topn
# ------------------------------------------------------------------------------------------------
PAR_LOG = []


def recording_dot_topn(indexes, P, Q, Qb, out_keys, out_scores, pool, k, num_threads=0):
    """Stands where parallel._core.dot_topn stands: logs what it is given and fills the outputs with a fixed pattern (some -1)."""
    PAR_LOG.append({"indexes": digest(indexes), "P": digest(P), "Q": digest(Q), "Qb": digest(Qb),
                    "pool": {"dtype": str(np.asarray(pool).dtype), "values": np.asarray(pool).tolist()}, "k": int(k),
                    "out_keys": [str(out_keys.dtype), list(out_keys.shape)], "out_scores": [str(out_scores.dtype), list(out_scores.shape)],
                    "same_matrix": bool(P is Q)})
    for r, idx in enumerate(np.asarray(indexes)):
        for c in range(out_keys.shape[1]):
            out_keys[r, c] = -1 if (int(idx) + c) % 5 == 4 else (int(idx) * 7 + c * 3) % Q.shape[0]
            out_scores[r, c] = 1.0 / (1 + r + c)


def par_calls(par_als, par_bpr, userkeys, itemkeys):
    """The same sequence of Par* calls for the reference's classes and for buffalo_amd.parallel's; returns a JSON-able log."""
    out = []

    def run(label, fn, *a, **k):
        del PAR_LOG[:]
        try:
            ret = fn(*a, **k)
            ret = [r.tolist() if isinstance(r, np.ndarray) else r for r in ret]
        except (RuntimeError, ValueError) as e:
            ret = "%s: %s" % (type(e).__name__, e)
        out.append({"call": label, "dot_topn": list(PAR_LOG), "returned": json.loads(json.dumps(ret))})
    some_users, some_items = userkeys[3:9] + ["nobody"] + userkeys[20:22], itemkeys[1:6] + ["nothing"] + itemkeys[10:12]
    run("als.topk_recommendation", par_als.topk_recommendation, some_users, topk=5)
    run("als.topk_recommendation pool=list repr", par_als.topk_recommendation, userkeys[:4], topk=4, pool=itemkeys[2:20] + ["nothing"], repr=True)
    run("als.topk_recommendation pool=ndarray", par_als.topk_recommendation, userkeys[5:8], topk=3, pool=np.array([1, 4, 9, 16], dtype=np.int32))
    run("als.topk_recommendation empty pool", par_als.topk_recommendation, userkeys[:2], topk=3, pool=["nothing"])
    run("bpr.topk_recommendation repr", par_bpr.topk_recommendation, some_users, topk=6, repr=True)
    run("als.most_similar item", par_als.most_similar, some_items, topk=5)
    run("als.most_similar user repr", par_als.most_similar, userkeys[2:5], topk=4, group="user", repr=True)
    run("als.most_similar pool", par_als.most_similar, itemkeys[:3], topk=3, pool=itemkeys[5:15])
    run("als.topk_recommendation after normalize", par_als.topk_recommendation, userkeys[:2], topk=3)
    return out


def par_models(ALS, BPRMF, als_opt, bpr_opt, make_data):
    """One ALS and one BPRMF model object (any front: the reference's or the stand-in's) with seeded factors and id maps."""
    rng = np.random.default_rng(33)
    U, I, d = 30, 24, 8
    userkeys, itemkeys = ["u%02d" % i for i in range(U)], ["i%02d" % i for i in range(I)]
    models = []
    for cls, opt in ((ALS, als_opt), (BPRMF, bpr_opt)):
        opt.update(dict(d=d, accelerator=True, num_workers=3))
        Recorder.trace = []
        m = cls(opt, data=make_data(U, I))
        m.P = rng.normal(size=(U, d)).astype(np.float32)
        m.Q = rng.normal(size=(I, d)).astype(np.float32)
        m.Qb = rng.normal(size=(I, 1)).astype(np.float32)
        m._idmanager.userids, m._idmanager.userid_map, m._idmanager.userid_mapped = userkeys, {k: i for i, k in enumerate(userkeys)}, True
        m._idmanager.itemids, m._idmanager.itemid_map, m._idmanager.itemid_mapped = itemkeys, {k: i for i, k in enumerate(itemkeys)}, True
        models.append(m)
    return models[0], models[1], userkeys, itemkeys


def reference_par():
    import buffalo.parallel.base as pb
    from buffalo.algo.als import ALS
    from buffalo.algo.bpr import BPRMF
    from buffalo.algo.options import ALSOption, BPRMFOption
    pb.dot_topn = recording_dot_topn
    als, bpr, userkeys, itemkeys = par_models(ALS, BPRMF, ALSOption().get_default_option(), BPRMFOption().get_default_option(),
                                             lambda U, I: None)
    return par_calls(pb.ParALS(als), pb.ParBPRMF(bpr), userkeys, itemkeys)


MODEL_FILES = {"bpr": os.path.join(HERE, "model_saved_by_reference_bprmf.bin"), "als": os.path.join(HERE, "model_saved_by_reference_als.bin")}


def reference_model_files():
    """Serializable.save (algo/base.py:275-294) of the reference's own BPRMF / ALS objects, seeded factors and id maps."""
    from buffalo.algo.als import ALS
    from buffalo.algo.bpr import BPRMF
    from buffalo.algo.options import ALSOption, BPRMFOption
    als, bpr, _, _ = par_models(ALS, BPRMF, ALSOption().get_default_option(), BPRMFOption().get_default_option(), lambda U, I: None)
    bpr.save(MODEL_FILES["bpr"])
    als.save(MODEL_FILES["als"])
    return {k: {"bytes": os.path.getsize(v), "crc32": zlib.crc32(open(v, "rb").read())} for k, v in MODEL_FILES.items()}


def main():
    install_reference()
    out = {name: reference_trace(name) for name in CASES}
    out.update({name: reference_trace_more(name) for name in MORE_CASES})
    out["validation_metrics"] = reference_metrics()
    out["parallel"] = reference_par()
    out["model_files"] = reference_model_files()
    json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
    for name, t in out.items():
        if "trace" in t:
            print(name, len(t["trace"]), "calls:", " ".join(c["call"] for c in t["trace"])[:400])
        else:
            print(name, str(t)[:600])


if __name__ == "__main__":
    main()
