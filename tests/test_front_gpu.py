"""End-to-end through the buffalo-compatible front: ALS / BPRMF / WARP `initialize(); train()` on planted
data with a validation split -- the template of /root/reference/tests/algo/base.py:56-97
(`_test3_init`, `_test4_train`, `_test5_validation`: ndcg / map thresholds)."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _data(seed=3):
    from buffalo_amd import synth
    from buffalo_front.data import MatrixMarketOptions
    csr, vali = synth.planted(500, 300, d_true=6, density=0.06, seed=seed)
    M = sp.csr_matrix((csr.vals, csr.keys, np.concatenate([[0], csr.indptr])), shape=(500, 300)).tolil()
    for u, i in vali:                      # put the held-out interactions back: the loader splits them off itself
        M[int(u), int(i)] = 1.0
    opt = MatrixMarketOptions().get_default_option()
    opt.input.main = M.tocsr()
    opt.data.validation = {"name": "sample", "p": 0.02, "max_samples": 400}
    return opt


@pytest.mark.parametrize("algo", ["ALS", "BPRMF", "WARP"])
def test_train_and_validate(algo):
    import buffalo_front.algo as A
    np.random.seed(7)
    data_opt = _data()
    if algo == "ALS":
        opt = A.ALSOption().get_default_option()
        opt.update(d=20, num_iters=8, validation={"topk": 10}, random_seed=7)
        m = A.ALS(opt, data_opt=data_opt)
    elif algo == "BPRMF":
        opt = A.BPRMFOption().get_default_option()
        opt.update(d=20, num_iters=60, lr=0.05, min_lr=0.01, reg_u=0.01, reg_i=0.01, reg_j=0.01, reg_b=0.01,
                   validation={"topk": 10}, evaluation_period=20, random_seed=7)
        m = A.BPRMF(opt, data_opt=data_opt)
    else:
        opt = A.WARPOption().get_default_option()
        opt.update(d=20, num_iters=20, validation={"topk": 10}, evaluation_period=5, random_seed=7)
        m = A.WARP(opt, data_opt=data_opt)
    m.initialize()
    header = m.data.get_header()
    assert m.P.shape == (header["num_users"], m.obj.get_vdim() if algo == "ALS" else 20)   # _test3_init
    calls = []
    ret = m.train(training_callback=lambda i, metrics: calls.append(i))                     # _test4 / _test5_1
    assert calls and "train_loss" in ret
    assert m.P.shape[1] == 20 and m.Q.shape[1] == 20 and np.isfinite(m.P).all() and np.isfinite(m.Q).all()
    res = m.get_validation_results()
    # a random ranking of 300 items scores ndcg@10 ~ 0.015: the planted structure must be found
    assert res["ndcg"] > 0.06 and res["map"] > 0.03, res                                    # _test5_validation
    top = m.topk_recommendation([0, 1, 2], topk=5)
    assert len(top) == 3 and all(len(v) == 5 for v in top.values())


@pytest.mark.parametrize("algo", ["ALS", "BPRMF"])
def test_train_and_validate_through_the_compiled_binding(algo, monkeypatch):
    """The same front run with `self.obj` = the COMPILED Cython class of integration/buffalo/algo/hip instead of the
    ctypes mirror: what stock buffalo does after the two-line change of INTEGRATION.md section 3.  The two bindings drive one library: same
    seeds, same arrays in -> the same model out, bit for bit."""
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "integration"))
    import build_binding
    if not os.path.exists(os.path.join(os.path.dirname(here), "integration", "buffalo", "algo", "hip", "_bpr.pyx")):
        pytest.skip("integration/ is not in this snapshot")
    try:
        build_binding.build()
    except ImportError as e:        # no Cython on this box and no prebuilt extension
        pytest.skip("the compiled binding is not built here (%s)" % e)
    CyBPR, CyALS, _ = build_binding.import_binding()
    import buffalo_front.algo as A
    import buffalo_front.algo.als as als_mod
    import buffalo_front.algo.bpr as bpr_mod

    def run(compiled):
        np.random.seed(7)
        if algo == "ALS":
            opt = A.ALSOption().get_default_option()
            opt.update(d=20, num_iters=4, validation={"topk": 10}, random_seed=7)
            if compiled:
                monkeypatch.setattr(als_mod, "CyALS", CyALS)
            m = A.ALS(opt, data_opt=_data())
        else:
            opt = A.BPRMFOption().get_default_option()
            opt.update(d=20, num_iters=20, lr=0.05, min_lr=0.01, reg_u=0.01, reg_i=0.01, reg_j=0.01, reg_b=0.01, validation={"topk": 10},
                       evaluation_period=10, random_seed=7)
            if compiled:
                monkeypatch.setattr(bpr_mod, "CyBPR", CyBPR)
            m = A.BPRMF(opt, data_opt=_data())
        if algo == "BPRMF":
            m.obj.set_mode("sequential", 1)         # the deterministic walk: two runs are comparable number for number
        m.initialize()
        m.train()
        assert type(m.obj).__module__ == ("buffalo.algo.hip._%s" % ("als" if algo == "ALS" else "bpr") if compiled else "buffalo_amd.backend")
        res = m.get_validation_results()
        monkeypatch.undo()
        return m.P.copy(), m.Q.copy(), res
    Pc, Qc, res_c = run(True)
    Pm, Qm, res_m = run(False)
    assert np.isfinite(Pc).all() and res_c["ndcg"] > 0.03, res_c
    np.testing.assert_array_equal(Pc, Pm)
    np.testing.assert_array_equal(Qc, Qm)
    assert res_c == res_m


def test_accelerator_false_is_refused():
    import buffalo_front.algo as A
    opt = A.BPRMFOption().get_default_option()
    opt.accelerator = False
    with pytest.raises(NotImplementedError):
        A.BPRMF(opt)


def test_save_load_roundtrip(tmp_path):
    import buffalo_front.algo as A
    np.random.seed(3)
    opt = A.ALSOption().get_default_option()
    opt.update(d=8, num_iters=1)
    m = A.ALS(opt, data_opt=_data())
    m.initialize()
    m.train()
    path = str(tmp_path / "als.bin")
    m.save(path)
    m2 = A.ALS(opt)
    m2.load(path)
    np.testing.assert_array_equal(m.P, m2.P)
    np.testing.assert_array_equal(m.Q, m2.Q)


def test_cfr_front_trains_on_the_device_like_on_the_oracle(tmp_path, oracle, monkeypatch):
    """Stream file -> loader (CSR groups and the `sppmi` group built on the device) -> CFR front -> `CyCFR` on the HIP backend,
    against the very same front run over the oracle's `OracleCFR` (tests/test_front_cfr_eals_cpu.py runs that side on the CPU).
    Tolerances are those of the backend-level parity (tests/test_cfr_gpu.py): factors 2e-3 of the largest entry, loss 1e-3."""
    import buffalo_front.algo.cfr as hc
    import test_front_cfr_eals_cpu as T
    from buffalo_front.algo.cfr import CFR
    data, _, _ = T._stream_data(tmp_path)
    sp_ = data.get_group("sppmi")
    assert len(sp_["key"]) > 0 and sp_["indptr"][-1] == len(sp_["key"])
    np.random.seed(11)
    m = CFR(T._cfr_opt(num_iters=3), data=data)
    m.initialize()
    ret = m.train()
    with monkeypatch.context() as mp:
        mp.setattr(hc, "CyCFR", oracle.OracleCFR)
        np.random.seed(11)
        o = CFR(T._cfr_opt(num_iters=3), data=data)
        o.initialize()
        want = o.train()
    assert abs(ret["train_loss"] - want["train_loss"]) <= 1e-3 * abs(want["train_loss"]), (ret, want)
    for a in ("U", "I", "C"):
        got, ref = getattr(m, a), getattr(o, a)
        assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max(), a
    top = m.topk_recommendation([0, 1, 2], topk=5)
    assert len(top) == 3 and all(len(v) == 5 for v in top.values())


def test_eals_front_trains_on_the_device_like_on_the_oracle(oracle, monkeypatch):
    import buffalo_front.algo.eals as he
    import test_front_cfr_eals_cpu as T
    from buffalo_front.algo.eals import EALS
    from buffalo_front.algo.options import EALSOption
    data = T._mm_data()
    opt = EALSOption().get_default_option()
    opt.update(d=10, num_iters=4, random_seed=3, validation={}, c0=64.0, exponent=0.5)
    np.random.seed(4)
    m = EALS(opt, data=data)
    m.initialize()
    ret = m.train()
    with monkeypatch.context() as mp:
        mp.setattr(he, "CyEALS", oracle.OracleEALS)
        np.random.seed(4)
        o = EALS(opt, data=data)
        o.initialize()
        want = o.train()
    assert abs(ret["train_loss"] - want["train_loss"]) <= 1e-3 * abs(want["train_loss"]), (ret, want)
    assert np.abs(m.P - o.P).max() <= 2e-3 * np.abs(o.P).max() and np.abs(m.Q - o.Q).max() <= 2e-3 * np.abs(o.Q).max()
