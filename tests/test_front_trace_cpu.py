"""The drop-in boundary as stock buffalo drives it: call traces recorded with the REFERENCE's own Python fronts
(tests/golden/make_front_traces.py: buffalo/algo/bpr.py and als.py imported unmodified from /root/reference, running over an
in-memory matrix against a recording stand-in for CuBPR / CuALS) against the same cases run through the stand-in front of
tests/front_harness.  A replacement backend sees exactly these calls: option file content, model binding (array dtypes,
shapes, contiguity, contents), placeholder, chunk boundaries of BufferedDataMatrix, loss samples, call order.

* `test_harness_front_issues_the_reference_call_sequence`: harness trace == committed golden trace, call by call;
* `test_golden_traces_are_current`: where /root/reference exists, the traces are regenerated and must equal the committed file
  (skipped on the GPU box, which has no reference tree).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_front_traces as G  # noqa: E402

GOLDEN = json.load(open(G.OUT))


def _harness_trace(name, monkeypatch):
    import buffalo_front.algo.als as ha
    import buffalo_front.algo.bpr as hb
    import buffalo_front.algo.warp as hw
    from buffalo_front.algo.options import ALSOption, BPRMFOption, WARPOption
    from buffalo_front.data import Data, MatrixMarketOptions
    monkeypatch.setattr(hb, "CyBPR", G.Recorder)
    monkeypatch.setattr(ha, "CyALS", G.Recorder)
    monkeypatch.setattr(hw, "CyWARP", G.Recorder)
    algo, shape, batch_mb, over = G.CASES[name]
    U, I, rows, cols, vals = G.case_matrix(*shape)
    dopt = MatrixMarketOptions().get_default_option()
    dopt.data.batch_mb = batch_mb
    data = Data(dopt)
    data.groups = G.groups_of(U, I, rows, cols, vals)
    data.header = {"num_nnz": len(rows), "num_users": U, "num_items": I, "completed": 1}
    opt = {"bpr": BPRMFOption, "als": ALSOption, "warp": WARPOption}[algo]().get_default_option()
    opt.update(over)
    opt.update(dict(accelerator=True, validation={}, evaluation_on_learning=False, save_best=False, num_workers=2))
    G.Recorder.trace = []
    model = {"bpr": hb.BPRMF, "als": ha.ALS, "warp": hw.WARP}[algo](opt, data=data)
    model.initialize()
    ret = model.train()
    return {"trace": G.Recorder.trace, "train_returned": {k: float(v) for k, v in ret.items()},
            "final_shapes": {k: list(getattr(model, k).shape) for k in ("P", "Q")}}


@pytest.mark.parametrize("name", sorted(G.CASES))
def test_harness_front_issues_the_reference_call_sequence(name, monkeypatch):
    got = json.loads(json.dumps(_harness_trace(name, monkeypatch), sort_keys=True))
    want = json.loads(json.dumps(GOLDEN[name]))
    if G.CASES[name][0] == "warp":   # the reference object had to be built with accelerator = False (see the case's note): the option file says so
        assert want["trace"][0]["args"][0]["option_file"]["accelerator"] is False
        want["trace"][0]["args"][0]["option_file"]["accelerator"] = True
    calls_got, calls_want = [c["call"] for c in got["trace"]], [c["call"] for c in want["trace"]]
    assert calls_got == calls_want, "call order differs:\n got  %s\n want %s" % (" ".join(calls_got), " ".join(calls_want))
    for i, (g, w) in enumerate(zip(got["trace"], want["trace"])):
        assert g == w, "call %d (%s) differs:\n got  %s\n want %s" % (i, w["call"], json.dumps(g)[:2000], json.dumps(w)[:2000])
    assert got["train_returned"] == want["train_returned"] and got["final_shapes"] == want["final_shapes"]


@pytest.mark.parametrize("name", sorted(G.MORE_CASES))
def test_cfr_and_eals_fronts_issue_the_reference_call_sequence(name, monkeypatch):
    """The same for the CFR and EALS fronts (f.4): set_embedding / precompute / partial_update_{user,item,context} with the row
    ranges of BufferedDataMatrix.fetch_batch_range, and initialize_model (negative weights) / precompute_cache / update /
    estimate_loss."""
    import buffalo_front.algo.cfr as hc
    import buffalo_front.algo.eals as he
    from buffalo_front.algo.options import CFROption, EALSOption
    from buffalo_front.data import Data, MatrixMarketOptions, StreamOptions
    monkeypatch.setattr(hc, "CyCFR", G.Recorder)
    monkeypatch.setattr(he, "CyEALS", G.Recorder)
    algo, shape, batch_mb, over = G.MORE_CASES[name]
    U, I, rows, cols, vals = G.case_matrix(*shape)
    dopt = (StreamOptions if algo == "cfr" else MatrixMarketOptions)().get_default_option()
    dopt.data.batch_mb = batch_mb
    data = Data(dopt)
    data.data_type = "stream" if algo == "cfr" else "matrix"
    data.groups = G.groups_of(U, I, rows, cols, vals)
    if algo == "cfr":
        data.groups["sppmi"] = G.sppmi_like(I, shape[3])
    data.header = {"num_nnz": len(rows), "num_users": U, "num_items": I, "completed": 1}
    opt = (CFROption if algo == "cfr" else EALSOption)().get_default_option()
    opt.update(over)
    opt.update(dict(validation={}, evaluation_on_learning=False, save_best=False))
    G.Recorder.trace = []
    model = (hc.CFR if algo == "cfr" else he.EALS)(opt, data=data)
    model.initialize()
    ret = model.train()
    got = json.loads(json.dumps({"trace": G.Recorder.trace, "train_returned": {k: float(v) for k, v in ret.items()}}, sort_keys=True))
    want = GOLDEN[name]
    calls_got, calls_want = [c["call"] for c in got["trace"]], [c["call"] for c in want["trace"]]
    assert calls_got == calls_want, "call order differs:\n got  %s\n want %s" % (" ".join(calls_got), " ".join(calls_want))
    for i, (g, w) in enumerate(zip(got["trace"], want["trace"])):
        assert g == w, "call %d (%s) differs:\n got  %s\n want %s" % (i, w["call"], json.dumps(g)[:2000], json.dumps(w)[:2000])
    assert got["train_returned"] == want["train_returned"]


def test_every_recorded_call_fits_the_ctypes_mirror():
    """Every call stock buffalo's fronts make on their Cython classes (all recorded traces) binds to a method of the matching
    class of buffalo_amd.backend -- the ctypes mirror of the C ABI -- with the recorded number of positional arguments."""
    import inspect

    from buffalo_amd import backend
    mirror = {"bpr": backend.CyBPR, "als": backend.CyALS, "warp": backend.CyWARP, "cfr": backend.CyCFR, "eals": backend.CyEALS}
    seen = set()
    for name, (algo, *_rest) in list(G.CASES.items()) + list(G.MORE_CASES.items()):
        cls = mirror[algo]
        for c in GOLDEN[name]["trace"]:
            fn = getattr(cls, c["call"], None)
            assert fn is not None, "%s has no method %s" % (cls.__name__, c["call"])
            inspect.signature(fn).bind(None, *c["args"])       # raises TypeError when the arity does not fit
            seen.add((algo, c["call"]))
    assert {("bpr", "add_jobs"), ("als", "partial_update"), ("cfr", "partial_update_item"), ("eals", "estimate_loss")} <= seen


def test_validation_metrics_match_the_reference_evaluation_code(monkeypatch):
    """NDCG / MAP / accuracy / AUC / RMSE / error computed by the reference's own Evaluable (evaluate/base.py:44-148, run by the
    generator over seeded factors and a held-out group) against the stand-in front's evaluation and the NDCG helper the GPU
    statistical tests use (tests/helpers.py:ndcg_at_k)."""
    import buffalo_front.algo.als as ha
    import helpers as H
    from buffalo_amd import synth
    from buffalo_front.algo.options import ALSOption
    from buffalo_front.data import Data, MatrixMarketOptions
    monkeypatch.setattr(ha, "CyALS", G.Recorder)
    U, I, rows, cols, vals, vali, P, Q = G.metrics_case()
    data = Data(MatrixMarketOptions().get_default_option())
    data.groups = dict(G.groups_of(U, I, rows, cols, vals), vali=vali)
    data.header = {"num_nnz": len(rows), "num_users": U, "num_items": I, "completed": 1}
    opt = ALSOption().get_default_option()
    opt.update(dict(d=20, accelerator=True, num_workers=1, validation={"topk": 10, "batch": 16, "eval_samples": 0}))
    G.Recorder.trace = []
    model = ha.ALS(opt, data=data)
    model.initialize()
    model.P, model.Q = P.copy(), Q.copy()

    def ranked(rows_, topk, pool=None):     # numpy scores + stable argsort: what the generator's quickselect stand-in does
        sc = model.P[np.asarray(rows_)] @ model.Q.T
        return [(int(r), np.argsort(-s, kind="stable")[:topk]) for r, s in zip(rows_, sc)]
    monkeypatch.setattr(model, "_get_topk_recommendation", ranked)
    for topk in (10, 25):
        want = GOLDEN["validation_metrics"]["topk%d" % topk]
        got = model.get_validation_results(topk=topk)
        assert set(got) == set(want)
        for k in want:
            assert abs(got[k] - want[k]) <= 1e-6 * max(1.0, abs(want[k])), (topk, k, got[k], want[k])
    g = data.groups["rowwise"]
    train = synth.CSR(U, I, g["indptr"], g["key"], g["val"])
    ndcg = H.ndcg_at_k(P, Q, train, list(zip(vali["row"].tolist(), vali["col"].tolist())), k=10)
    assert abs(ndcg - GOLDEN["validation_metrics"]["topk10"]["ndcg"]) < 1e-9


def test_par_classes_match_the_reference_par_classes(monkeypatch):
    """The harness's ParALS / ParBPRMF stand-ins against the reference's parallel/base.py:77-156, both around the same
    recording dot_topn: what reaches dot_topn (indices, factor matrices, bias, pool, k, output buffers, "P is Q") and what comes
    back to the caller (kept keys -- incl. the reference's slip when a key is unknown --, index / key lists with -1 dropped
    under repr, scores, the errors for an empty pool and for normalised factors)."""
    import buffalo_front.algo.als as ha
    import buffalo_front.algo.bpr as hb
    from buffalo_front import parallel as par
    from buffalo_front.algo.options import ALSOption, BPRMFOption
    from buffalo_front.data import Data, MatrixMarketOptions
    monkeypatch.setattr(ha, "CyALS", G.Recorder)
    monkeypatch.setattr(hb, "CyBPR", G.Recorder)
    monkeypatch.setattr(par, "dot_topn", G.recording_dot_topn)

    def make_data(U, I):
        d = Data(MatrixMarketOptions().get_default_option())
        d.header = {"num_nnz": 0, "num_users": U, "num_items": I, "completed": 1}
        return d
    als, bpr, userkeys, itemkeys = G.par_models(ha.ALS, hb.BPRMF, ALSOption().get_default_option(), BPRMFOption().get_default_option(), make_data)
    got = G.par_calls(par.ParALS(als), par.ParBPRMF(bpr), userkeys, itemkeys)
    want = GOLDEN["parallel"]
    assert [c["call"] for c in got] == [c["call"] for c in want]
    for g, w in zip(got, want):
        for gd, wd in zip(g["dot_topn"], w["dot_topn"]):
            # The one deliberate difference: a pool given as a LIST of keys reaches the reference's dot_topn as the int64 array
            # np.array([...]) makes of it (algo/base.py:262), which the typed buffer of _core.pyx:45 (np.int32_t) rejects --
            # stock buffalo only works with int32 ndarray pools (its own tests use those).  buffalo_amd hands over int32.
            assert gd["pool"]["dtype"] == "int32" and wd["pool"]["dtype"] in ("int32", "int64")
            gd["pool"]["dtype"] = wd["pool"]["dtype"]
        assert g["dot_topn"] == w["dot_topn"], "%s: dot_topn saw\n %s\nwant\n %s" % (w["call"], json.dumps(g["dot_topn"])[:1500], json.dumps(w["dot_topn"])[:1500])
        assert g["returned"] == w["returned"], "%s returned\n %s\nwant\n %s" % (w["call"], json.dumps(g["returned"])[:1500], json.dumps(w["returned"])[:1500])


def test_model_files_saved_by_the_reference_classes(monkeypatch, tmp_path):
    """tests/golden/model_saved_by_reference_{bprmf,als}.bin were written by `Serializable.save` of the reference's own BPRMF /
    ALS objects (algo/base.py:275-294).  buffalo_amd.serialize reads them, writes the same content back byte for byte, and the
    stand-in models built the same way save to identical files (pickled Option class path included)."""
    import buffalo_front.algo.als as ha
    import buffalo_front.algo.bpr as hb
    from buffalo_amd.serialize import Option, dump_objects, load_objects
    from buffalo_front.algo.options import ALSOption, BPRMFOption
    from buffalo_front.data import Data, MatrixMarketOptions
    monkeypatch.setattr(ha, "CyALS", G.Recorder)
    monkeypatch.setattr(hb, "CyBPR", G.Recorder)

    def make_data(U, I):
        d = Data(MatrixMarketOptions().get_default_option())
        d.header = {"num_nnz": 0, "num_users": U, "num_items": I, "completed": 1}
        return d
    als, bpr, userkeys, itemkeys = G.par_models(ha.ALS, hb.BPRMF, ALSOption().get_default_option(), BPRMFOption().get_default_option(), make_data)
    for kind, model, fields in (("bpr", bpr, ["_idmanager", "opt", "Q", "Qb", "P"]), ("als", als, None)):
        golden = open(G.MODEL_FILES[kind], "rb").read()
        objs = load_objects(G.MODEL_FILES[kind])
        names = [n for n, _ in objs]
        if fields:
            assert names == fields
        d = dict(objs)
        assert isinstance(d["opt"], Option) and d["opt"].d == 8 and d["_idmanager"].itemids == itemkeys and d["_idmanager"].userid_map["u07"] == 7
        assert np.array_equal(d["P"], model.P) and np.array_equal(d["Q"], model.Q)
        back = tmp_path / ("%s_back.bin" % kind)
        dump_objects(str(back), objs)
        assert back.read_bytes() == golden
        ours = tmp_path / ("%s_ours.bin" % kind)
        model.save(str(ours))
        assert ours.read_bytes() == golden, kind


@pytest.mark.skipif(not os.path.isdir(G.REF), reason="the reference tree is not on this machine")
def test_golden_traces_are_current(tmp_path):
    """Regenerate with the reference's fronts (a fresh interpreter: the generator installs stub modules) and compare."""
    out = tmp_path / "traces.json"
    code = "import sys; sys.path.insert(0, %r); import make_front_traces as G; G.OUT = %r; G.main()" % (os.path.join(HERE, "golden"), str(out))
    code = code.replace("G.main()", "G.MODEL_FILES = {k: %r + k + '.bin' for k in G.MODEL_FILES}; G.main()" % (str(tmp_path) + os.sep))
    subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, cwd=os.path.dirname(HERE))
    assert json.load(open(out)) == GOLDEN
    for k, path in G.MODEL_FILES.items():
        assert open(path, "rb").read() == open(str(tmp_path / (k + ".bin")), "rb").read()


@pytest.mark.skipif(not os.path.isdir("/root/reference/buffalo"), reason="/root/reference is not here")
def test_no_cython_drop_in_reaches_the_library_from_stock_buffalo_s_fronts():
    """with `buffalo.algo.cuda._bpr.CyBPR` / `_als.CyALS` bound to `buffalo_amd.backend`'s classes, stock
    buffalo's unmodified `ALS` / `BPRMF` with accelerator = True construct the product's handle.  Without a GPU that must fail
    loudly inside libbuffalo_hip (no CPU fallback) -- on a GPU box the same binding trains."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "golden", "run_reference_tests.py"), "dropin"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("dropin ")]
    assert len(lines) == 2
    import torch
    for l in lines:
        if torch.cuda.is_available():
            assert "constructed over buffalo_amd.backend.Cy" in l, l
        else:
            assert "BuffaloHipError: no HIP device available" in l, l
