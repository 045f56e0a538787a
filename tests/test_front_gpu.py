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
