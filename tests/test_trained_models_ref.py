"""Whole training runs: stock buffalo's Python over the oracle (golden, tests/golden/make_trained_models.py) against the stand-in
fronts from the same files and the same seeds.

The golden side is the reference's own `ALS` / `EALS` front in CPU mode, its MatrixMarket loader and its evaluation, unmodified,
with the oracle's classes as its compiled backends.  On CPU the stand-in front runs over the same oracle: validation draw, initial
factors, every epoch and every metric must come out IDENTICAL -- the stand-in is the reference's Python, result for result.  On a
GPU box the same front drives the HIP backend: the trained factors must agree with the golden ones within the kernels' parity
tolerance after 3-4 free-running epochs, and the validation metrics (ranked on the device) with them."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_trained_models as mk  # noqa: E402
import helpers as H  # noqa: E402

GOLDEN = np.load(mk.OUT)
META = json.loads(str(GOLDEN["meta"]))


def _train(tmp_path, name, modes=None):
    import buffalo_front.algo as A
    from buffalo_front.data import MatrixMarketOptions
    algo, shape, over, np_seed = mk.CASES[name]
    cls, opt_cls = {"als": (A.ALS, A.ALSOption), "eals": (A.EALS, A.EALSOption), "bpr": (A.BPRMF, A.BPRMFOption),
                    "warp": (A.WARP, A.WARPOption)}[algo]
    path = tmp_path / "main.mtx"
    path.write_text(mk.coordinate_text(*shape))
    opt = opt_cls().get_default_option()
    opt.update(over)
    np.random.seed(np_seed)
    model = cls(opt, data_opt=mk.data_option(MatrixMarketOptions, str(path)))
    for k, v in (modes or {}).items():
        model.obj.set_mode(k, v)
    model.initialize()
    ret = model.train()
    return model, ret, model.get_validation_results()


@pytest.mark.parametrize("name", sorted(mk.CASES))
def test_stand_in_front_over_the_oracle_equals_stock_buffalo_over_the_oracle(tmp_path, oracle, monkeypatch, name):
    import buffalo_front.algo.als as ha
    import buffalo_front.algo.base as hb
    import buffalo_front.algo.bpr as hp
    import buffalo_front.algo.eals as he
    import buffalo_front.algo.warp as hw
    import buffalo_front.data as D

    class OracleBehindTheAcceleratorSurface(oracle.OracleALS):
        """CuALS's extra calls (cuda/_als.pyx:35-67) on the CPU class: no padding, nothing to announce."""
        def init(self, opt_path):
            with open(opt_path) as f:
                self._d = json.load(f)["d"]
            return super().init(opt_path)

        def get_vdim(self):
            return self._d

        def set_placeholder(self, *args):
            pass

    class OracleRanker:
        def dot_topn(self, rows, P, Q, Qb, keys, scores, pool, k):
            oracle.dot_topn(np.ascontiguousarray(rows, dtype=np.int32), P, Q, Qb, keys, scores, pool, k)
    monkeypatch.setattr(D, "_group", lambda nr, nc, r, c, v: oracle.coo_to_csr(r, c, v, nr, nc))
    monkeypatch.setattr(ha, "CyALS", OracleBehindTheAcceleratorSurface)
    monkeypatch.setattr(he, "CyEALS", oracle.OracleEALS)
    monkeypatch.setattr(hp, "CyBPR", mk.accelerator_over_oracle(oracle.OracleBPRMF))     # the very classes the golden run bound
    monkeypatch.setattr(hw, "CyWARP", mk.accelerator_over_oracle(oracle.OracleWARP))
    monkeypatch.setattr(hb.Algo, "_ranker", lambda self: OracleRanker())
    model, ret, vali = _train(tmp_path, name)
    want = META[name]
    assert model.data.get_header()["num_nnz"] == want["header"]["num_nnz"]
    assert np.array_equal(model.P, GOLDEN[name + "/P"]) and np.array_equal(model.Q, GOLDEN[name + "/Q"])
    if name + "/Qb" in GOLDEN.files:
        assert np.array_equal(model.Qb, GOLDEN[name + "/Qb"])
    assert ret["train_loss"] == want["train"]["train_loss"]
    rel = lambda k: 1e-9   # noqa: E731 -- rmse / error included: the stand-in keeps the reference's entry-by-entry float32 sums
    for k, v in want["validation"].items():
        assert abs(vali[k] - v) <= rel(k) * max(1.0, abs(v)), (k, vali[k], v)
    for k, v in want["train"].items():
        assert abs(ret[k] - v) <= rel(k) * max(1.0, abs(v)), (k, ret[k], v)


# what the device run is held to, per case: (backend modes, factor tolerance relative to the largest entry | "envelope", loose metrics?)
DEVICE = {
    "als_llt_d32": ({}, 5e-3, False), "eals_d16": ({}, 5e-3, False),
    # twelve CG steps, regulariser 5, d = 64 over 2,000 items: the oracle follows the float64 recurrence to 1e-4 through the whole
    # run (measured when the case was made), so the device has to reach the reference's model itself
    "als_manual_cg_d64_wellposed": ({}, 5e-3, False),
    # the reference's DEFAULT solver (three CG steps) from the |N(0, 1/d^2)| start over 90 items: the first user half-epoch returns
    # rows of size y / reg, the item systems that follow have condition numbers beyond fp32, and the oracle ITSELF ends several
    # per cent from the float64 evaluation of its own recurrence (scripts/als_cg_diag.py front; profiles/r03_als_cg_d64_diag.txt:
    # every single call of the HIP run is CLOSER to float64 than the oracle's replay of the same call, ratio 0.75 - 1.3).
    # Free-running fp32 runs of this case agree with each other no better than each agrees with float64 -- which is the bound:
    # Held call by call to the envelope of tests/test_als_gpu.py (every call replayed on the oracle and in float64 from the same
    # inputs); the free-running end state only to the measured amplification of such rounding on this case.
    "als_manual_cg_d64": ({}, "envelope", True),
    # the golden run used the oracle's deterministic modes; the backend's `sequential` walk follows the same sample stream and order
    # (tests/test_bpr_gpu.py: 1e-5 after 3 epochs), its frozen-epoch paths (adagrad, WARP) are order-free up to summation (1e-4)
    "bpr_sgd_d20": ({"sequential": 1}, 1e-4, False), "bpr_adagrad_d40": ({}, 1e-3, False), "warp_d24": ({}, 1e-3, False),
}


def _replaying_backend(records):
    """CyALS that replays every partial_update it is given on the oracle and on the float64 recurrence (tests/f64_backend.py) FROM
    THE SAME INPUTS -- the host factors right before the call -- and records the three distances over the updated rows."""
    from buffalo_amd.backend import CyALS
    from f64_backend import F64ALS
    from oracle import oracle as orc

    class Replaying(CyALS):
        def init(self, opt_path):
            self._path = opt_path.decode("utf-8") if isinstance(opt_path, bytes) else opt_path
            with open(self._path) as f:
                self._opt = json.load(f)
            return super().init(opt_path)

        def initialize_model(self, P, Q):
            self._F = (P, Q)
            return super().initialize_model(P, Q)

        def partial_update(self, start_x, next_x, indptr, keys, vals, axis):
            d = self._opt["d"]
            P0, Q0 = self._F[0][:, :d].copy(), self._F[1][:, :d].copy()
            out = super().partial_update(start_x, next_x, indptr, keys, vals, axis)
            got = self._F[axis][start_x:next_x, :d]
            o = orc.OracleALS()
            assert o.init(H.write_opt(dict(self._opt, accelerator=False)))
            Fo = (P0.copy(), Q0.copy())
            o.initialize_model(*Fo)
            o.precompute(axis)
            o.partial_update(start_x, next_x, indptr, keys, vals, axis)
            f = F64ALS()
            assert f.init(H.write_opt(dict(self._opt, accelerator=False)))
            f.initialize_model(P0.copy(), Q0.copy())
            f.precompute(axis)
            f.partial_update(start_x, next_x, indptr, keys, vals, axis)
            truth = f.F[axis][start_x:next_x]
            records.append((axis, H.relerr(got, truth), H.relerr(Fo[axis][start_x:next_x], truth), H.relerr(got, Fo[axis][start_x:next_x])))
            return out
    return Replaying


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(mk.CASES))
def test_stand_in_front_over_the_device_reaches_stock_buffalo_s_model(tmp_path, monkeypatch, name):
    """Same file, same seeds, the HIP backend: initial factors and validation split are identical by construction, the epochs run
    free.  Factors per DEVICE above; train loss 1e-3 (2e-2 loose), rmse / error 1e-3 (2e-2), ranking metrics 0.03 (0.06) -- with
    ~60 held-out entries one flipped rank moves accuracy by 0.017.  Measured on a device: profiles/r03_trained_models_device.txt.

    "envelope" (the three-step CG case): every call the front issues is replayed on the oracle and in float64 from the same inputs
    and held to the half-epoch tests' envelope, err(hip, f64) <= max(2.5 err(oracle, f64), 5e-5) -- the kernels add no more rounding
    than the reference's own arithmetic does, call by call; the free-running END state is only held to the spread such rounding
    is amplified to on this case (scripts/als_cg_diag.py; profiles/r03_als_cg_d64_diag.txt: a random perturbation of the size of
    the oracle's own rounding in the first item half-epoch, 7e-4, ends 0.7 - 3.3 % away from the unperturbed run; the oracle
    itself ends 1.9 % from the float64 run of the whole recurrence)."""
    modes, tol, loose = DEVICE[name]
    records = []
    if tol == "envelope":
        import buffalo_front.algo.als as ha
        monkeypatch.setattr(ha, "CyALS", _replaying_backend(records))
    model, ret, vali = _train(tmp_path, name, modes)
    want = META[name]
    for i, (axis, e_hip, e_or, e_pair) in enumerate(records):
        env = max(2.5 * e_or, 5e-5)
        print("\n%s call %d axis %d: err(hip, f64) %.3e  err(oracle, f64) %.3e  ratio %.2f  hip~oracle %.3e" % (name, i, axis, e_hip, e_or, e_hip / max(e_or, 1e-30), e_pair))
        assert e_hip <= env and e_pair <= 4 * env, (i, axis, e_hip, e_or, e_pair)
    for f in ("P", "Q", "Qb"):
        if "%s/%s" % (name, f) not in GOLDEN.files:
            continue
        got, ref = getattr(model, f), GOLDEN["%s/%s" % (name, f)]
        bound = 0.15 if tol == "envelope" else tol
        print("\n%s %s: hip~golden %.3e (bound %.1e)" % (name, f, np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30), bound))
        assert got.shape == ref.shape and np.abs(got - ref).max() <= bound * max(np.abs(ref).max(), 1e-30), (f, np.abs(got - ref).max(), np.abs(ref).max())
    assert abs(ret["train_loss"] - want["train"]["train_loss"]) <= (2e-2 if loose else 1e-3) * abs(want["train"]["train_loss"])
    for k, v in want["validation"].items():
        bound = (2e-2 if loose else 1e-3) * abs(v) if k in ("rmse", "error") else (0.06 if loose else 0.03)
        assert abs(vali[k] - v) <= bound, (k, vali[k], v)


@pytest.mark.skipif(not os.path.isdir("/root/reference/buffalo"), reason="/root/reference is not here")
def test_golden_models_are_what_stock_buffalo_trains_now(tmp_path):
    import subprocess
    out = str(tmp_path / "fresh.npz")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_trained_models.py"), "--out", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    fresh = np.load(out)
    assert sorted(fresh.files) == sorted(GOLDEN.files)
    for k in fresh.files:
        assert np.array_equal(fresh[k], GOLDEN[k]), k
