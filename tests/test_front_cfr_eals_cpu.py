"""The stand-in CFR / EALS fronts trained end to end on the CPU, with the oracle's `OracleCFR` / `OracleEALS` (the restatements of
lib/algo_impl/cfr/cfr.cc and eals/eals.hpp, same method surface as the Cython classes) where the HIP backend's `CyCFR` /
`CyEALS` stand on a GPU box -- tests/test_front_gpu.py runs the same flows on the device.

What is under test is everything the fronts and the loaders do around the backend: the Stream loader's `sppmi` group (pair
lines of the TRAINING part of every sequence, in sequence order), the three block sweeps of a CFR epoch over
`fetch_batch_range` / `get_specific_chunk`, eALS's negative weights and cache calls -- by comparing the front's result with
the same backend driven directly over the whole matrix, as tests/test_cfr_gpu.py / test_eals_gpu.py drive it."""
import numpy as np
import pytest

from oracle import ref_fileio as rf


@pytest.fixture(autouse=True)
def _oracle_everywhere(monkeypatch, oracle):
    import buffalo_front.algo.cfr as hc
    import buffalo_front.algo.eals as he
    import buffalo_front.data as D
    monkeypatch.setattr(D, "_group", lambda nr, nc, r, c, v: oracle.coo_to_csr(r, c, v, nr, nc))
    monkeypatch.setattr(D, "_sppmi_group", lambda ip, it, ni, w, k: {n: oracle.build_sppmi(ip, it, ni, w, k)[n] for n in ("indptr", "key", "val")})
    monkeypatch.setattr(hc, "CyCFR", oracle.OracleCFR)
    monkeypatch.setattr(he, "CyEALS", oracle.OracleEALS)


def _stream_files(tmp_path, num_users=120, num_items=60, seed=0):
    """Sequences with planted structure: every user draws from one of four item clusters (plus noise), 3..24 events."""
    rng = np.random.default_rng(seed)
    names = ["i%03d" % i for i in range(num_items)]
    seqs = []
    for u in range(num_users):
        c = u % 4
        n = int(rng.integers(3, 25))
        own = rng.integers(c * num_items // 4, (c + 1) * num_items // 4, size=n)
        noise = rng.integers(0, num_items, size=n)
        ids = np.where(rng.random(n) < 0.85, own, noise)
        seqs.append([names[i] for i in ids])
    (tmp_path / "main").write_text("".join(" ".join(s) + "\n" for s in seqs))
    (tmp_path / "uid").write_text("".join("u%d\n" % u for u in range(num_users)))
    (tmp_path / "iid").write_text("".join(n + "\n" for n in names))
    return seqs, names


def _stream_data(tmp_path, batch_mb=1024, vali=True, **kw):
    from buffalo_front.data import Stream, StreamOptions
    seqs, names = _stream_files(tmp_path, **kw)
    opt = StreamOptions().get_default_option()
    opt.input.main, opt.input.uid, opt.input.iid = str(tmp_path / "main"), str(tmp_path / "uid"), str(tmp_path / "iid")
    opt.data.internal_data_type = "matrix"                     # cfr.py:52
    opt.data.validation = {"name": "newest", "n": 1, "max_samples": 500} if vali else {}
    opt.data.sppmi = {"windows": 3, "k": 1}
    opt.data.batch_mb = batch_mb
    d = Stream(opt)
    d.create()
    return d, seqs, names


def test_stream_loader_builds_the_sppmi_group_from_the_training_sequences(tmp_path, oracle):
    d, seqs, names = _stream_data(tmp_path)
    index = {n: i for i, n in enumerate(names)}
    train = [[index[w] for w in s][:len(s) - 1] for s in seqs]           # `newest`, n = 1: the last event of every sequence is held out
    indptr = np.cumsum([len(t) for t in train]).astype(np.int64)
    items = np.concatenate([np.asarray(t, np.int32) for t in train])
    want = oracle.build_sppmi(indptr, items, len(names), 3, 1)
    g = d.get_group("sppmi")
    for n in ("indptr", "key", "val"):
        assert np.array_equal(g[n], want[n]), n
    assert d.get_header()["sppmi_nnz"] == len(want["key"]) > 0 and d.data_type == "stream"
    assert d.get_group("vali")["row"].shape[0] == len(seqs)
    if rf.available():                                                    # ... and what the reference's own compiled builder makes of the same lines
        ref = rf.canonical_rows(rf.build_sppmi(indptr, items, len(names), 3, 1, num_workers=2))
        for n in ("indptr", "key", "val"):
            assert np.array_equal(g[n], ref[n]), n


def _cfr_opt(**kw):
    from buffalo_front.algo.options import CFROption
    opt = CFROption().get_default_option()
    opt.update(d=12, num_iters=1, random_seed=5, compute_loss=True, validation={}, optimizer="llt", l=0.7, alpha=4.0)   # cfr.cc:47 reads `compute_loss`
    opt.update(**kw)
    return opt


def _drive_cfr_directly(oracle, opt, data, arrs, epochs):
    """tests/test_cfr_gpu.py:_epoch over whole groups: precompute(item) -> users, precompute(user) -> items, contexts."""
    import json
    import os
    import tempfile
    obj = oracle.OracleCFR()
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(dict(opt), f)
    assert obj.init(f.name)
    os.unlink(f.name)
    for name in ("user", "item", "context", "item_bias", "context_bias"):
        obj.set_embedding(arrs[name], name)
    rw, cw, sp_ = (data.get_group(g) for g in ("rowwise", "colwise", "sppmi"))
    U, I = len(rw["indptr"]), len(cw["indptr"])
    loss = 0.0
    for _ in range(epochs):
        obj.precompute("item")
        loss = obj.partial_update_user(0, U, rw["indptr"], rw["key"], rw["val"])
        obj.precompute("user")
        loss += obj.partial_update_item(0, I, cw["indptr"], cw["key"], cw["val"], sp_["indptr"], sp_["key"], sp_["val"])
        loss += obj.partial_update_context(0, I, sp_["indptr"], sp_["key"], sp_["val"])
    return loss


@pytest.mark.parametrize("batch_mb", [1024, 0.004])
def test_cfr_front_trains_and_equals_the_backend_driven_directly(tmp_path, oracle, batch_mb):
    from buffalo_front.algo.cfr import CFR
    data, _, _ = _stream_data(tmp_path, batch_mb=batch_mb)
    np.random.seed(11)
    m = CFR(_cfr_opt(num_iters=3), data=data)
    m.initialize()
    start = {n: getattr(m, a).copy() for a, n in (("U", "user"), ("I", "item"), ("C", "context"), ("Ib", "item_bias"), ("Cb", "context_bias"))}
    ret = m.train()
    assert np.isfinite(ret["train_loss"]) and ret["train_loss"] > 0
    # one epoch less must leave a larger loss: the sweeps do descend
    np.random.seed(11)
    m1 = CFR(_cfr_opt(num_iters=1), data=data)
    m1.initialize()
    assert m1.train()["train_loss"] > ret["train_loss"]
    if batch_mb == 1024:
        # the front's three sweeps in one range each == the backend driven over the whole groups (same calls, same order)
        direct = _drive_cfr_directly(oracle, m.opt, data, start, epochs=3)
        assert abs(direct / m.compute_scale() - ret["train_loss"]) <= 1e-6 * abs(ret["train_loss"])
        for a, n in (("U", "user"), ("I", "item"), ("C", "context")):
            assert np.array_equal(getattr(m, a), start[n])          # trained in place: the arrays handed over ARE the model
    else:
        # 4 KB of budget: several row ranges per sweep (buffered_data.py:122-158), the last row on its own is never fed (Q-24-like)
        from buffalo_front.data import BufferedDataMatrix
        buf = BufferedDataMatrix()
        buf.initialize(data, with_sppmi=True)
        assert len(list(buf.fetch_batch_range(["rowwise"]))) > 1 and len(list(buf.fetch_batch_range(["colwise", "sppmi"]))) > 1


def _mm_data(seed=2):
    import scipy.sparse as sp
    from buffalo_front.data import MatrixMarket, MatrixMarketOptions
    rng = np.random.default_rng(seed)
    U, I = 90, 50
    taste = rng.integers(0, 5, U)
    dense = (rng.random((U, I)) < np.where((np.arange(I)[None, :] % 5) == taste[:, None], 0.5, 0.03))
    M = sp.csr_matrix(dense.astype(np.float32) * rng.integers(1, 4, (U, I)))
    opt = MatrixMarketOptions().get_default_option()
    opt.input.main = M
    opt.data.validation = {}
    d = MatrixMarket(opt)
    d.create()
    return d


def test_eals_front_trains_and_equals_the_backend_driven_directly(oracle):
    import json
    import os
    import tempfile
    from buffalo_front.algo.eals import EALS
    from buffalo_front.algo.options import EALSOption
    data = _mm_data()
    opt = EALSOption().get_default_option()
    opt.update(d=10, num_iters=4, random_seed=3, validation={}, c0=64.0, exponent=0.5)
    np.random.seed(4)
    m = EALS(opt, data=data)
    m.initialize()
    P0, Q0, C0 = m.P.copy(), m.Q.copy(), m.C.copy()
    # eals.py:104-112
    cw = data.get_group("colwise")
    pop = np.diff(np.concatenate([[0], cw["indptr"]])).astype(np.float32)
    powered = (pop / pop.max()) ** 0.5
    np.testing.assert_allclose(C0, 64.0 * powered / powered.sum(), rtol=1e-6)
    ret = m.train()
    np.random.seed(4)
    m1 = EALS(dict(opt, num_iters=1), data=data)
    m1.initialize()
    assert m1.train()["train_loss"] > ret["train_loss"] > 0
    # the same backend driven directly
    obj = oracle.OracleEALS()
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(dict(m.opt), f)
    assert obj.init(f.name)
    os.unlink(f.name)
    P, Q = P0.copy(), Q0.copy()
    obj.initialize_model(P, Q, C0)
    rw = data.get_group("rowwise")
    nnz = data.get_header()["num_nnz"]
    obj.precompute_cache(nnz, rw["indptr"], rw["key"], 0)
    obj.precompute_cache(nnz, cw["indptr"], cw["key"], 1)
    for _ in range(4):
        assert obj.update(rw["indptr"], rw["key"], rw["val"], 0)
        assert obj.update(cw["indptr"], cw["key"], cw["val"], 1)
    loss, _ = obj.estimate_loss(nnz, rw["indptr"], rw["key"], rw["val"], 0)
    assert loss == ret["train_loss"]
    assert np.array_equal(P, m.P) and np.array_equal(Q, m.Q)
