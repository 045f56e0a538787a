"""Host-side plumbing that needs no GPU: option objects, in-memory MatrixMarket / Stream loaders and the
BufferedDataMatrix chunk semantics (incl. Q-24), mirrored from the reference's data tests
(/root/reference/tests/data/test_mm.py:14-22 builds tiny inline files the same way)."""
import numpy as np
import pytest
import scipy.sparse as sp

from buffalo_front.algo.options import ALSOption, BPRMFOption, WARPOption
from buffalo_front.data import BufferedDataMatrix, MatrixMarket, MatrixMarketOptions, Stream, StreamOptions, load
from buffalo_front.misc import Option


@pytest.fixture(autouse=True)
def _oracle_builds_the_groups(monkeypatch, oracle):
    """The loaders hand their (row, col, val) records to the device (`buffalo_amd.ingest`); there is no GPU in
    this suite, so the CPU oracle's restatement of the same step stands in for it.  What is under test here
    is the host logic around it (parsing, hold-out, chunking); tests/test_ingest_gpu.py runs the same loader
    checks against the real device path."""
    import buffalo_front.data as D
    monkeypatch.setattr(D, "_group", lambda nr, nc, r, c, v: oracle.coo_to_csr(r, c, v, nr, nc))


def test_option_defaults_and_validation():
    for cls, key, val in ((ALSOption, "alpha", 8.0), (BPRMFOption, "lr", 0.002), (WARPOption, "max_trials", 500)):
        o = cls()
        opt = o.get_default_option()
        assert opt[key] == val and isinstance(opt, Option)
        assert o.is_valid_option(opt)
        bad = Option(dict(opt))
        bad["d"] = "20"
        with pytest.raises(RuntimeError):
            o.is_valid_option(bad)
    with pytest.raises(RuntimeError):
        ALSOption().is_valid_option(Option(dict(ALSOption().get_default_option(), optimizer="sgd")))
    path = ALSOption().create_temporary_option_from_dict(ALSOption().get_default_option())
    from buffalo_front.misc import load_option
    assert load_option(path).d == 20


def _mm(tmp_path, validation=True):
    M = sp.random(40, 30, density=0.2, format="coo", random_state=3)
    M.data[:] = np.random.default_rng(0).integers(1, 5, size=M.nnz)
    import scipy.io
    p = tmp_path / "main.mtx"
    scipy.io.mmwrite(str(p), M)
    opt = MatrixMarketOptions().get_default_option()
    opt.input.main = str(p)
    if not validation:
        opt.data.validation = {}
    return M, opt


def test_matrix_market_layout(tmp_path):
    M, opt = _mm(tmp_path, validation=False)
    d = load(opt)
    d.create()
    h = d.get_header()
    assert (h["num_users"], h["num_items"], h["num_nnz"]) == (40, 30, M.nnz)
    rw, cw = d.get_group("rowwise"), d.get_group("colwise")
    assert rw["indptr"].dtype == np.int64 and rw["key"].dtype == np.int32 and rw["val"].dtype == np.float32
    assert rw["indptr"].shape == (40,) and rw["indptr"][-1] == M.nnz          # END offsets, no leading zero
    csr = M.tocsr()
    csr.sort_indices()
    np.testing.assert_array_equal(rw["indptr"], csr.indptr[1:])
    np.testing.assert_array_equal(rw["key"], csr.indices)                       # sorted ascending per row
    np.testing.assert_allclose(rw["val"], csr.data)
    csc = M.tocsc()
    csc.sort_indices()
    np.testing.assert_array_equal(cw["indptr"], csc.indptr[1:])
    np.testing.assert_array_equal(cw["key"], csc.indices)


def test_matrix_market_validation_holdout(tmp_path):
    M, opt = _mm(tmp_path)
    opt.data.validation = {"name": "sample", "p": 0.1, "max_samples": 7}
    np.random.seed(1)
    d = MatrixMarket(opt)
    d.create()
    v = d.get_group("vali")
    assert v["row"].shape[0] == 7 and d.get_header()["num_nnz"] == M.nnz - 7
    train = set(zip(d.get_group("rowwise")["key"], d.groups["rowwise"]["indptr"].searchsorted(np.arange(M.nnz - 7), side="right")))
    assert len(train) == M.nnz - 7


def test_stream_loader(tmp_path):
    (tmp_path / "main").write_text("a b a c\nb\nc c d a\n")
    (tmp_path / "uid").write_text("u0\nu1\nu2\n")
    (tmp_path / "iid").write_text("a\nb\nc\nd\n")
    opt = StreamOptions().get_default_option()
    opt.input.main, opt.input.uid, opt.input.iid = str(tmp_path / "main"), str(tmp_path / "uid"), str(tmp_path / "iid")
    opt.data.validation = {"name": "newest", "n": 1, "max_samples": 1}   # every held-out entry is kept (stream.py:100-118): 2 > max_samples
    d = Stream(opt)
    d.create()
    rw = d.get_group("rowwise")
    # u0: a b a | c held out ; u1: b (single item is never held out) ; u2: c c d | a held out
    np.testing.assert_array_equal(rw["indptr"], [2, 3, 5])
    np.testing.assert_array_equal(rw["key"], [0, 1, 1, 2, 3])
    np.testing.assert_allclose(rw["val"], [2, 1, 1, 2, 1])
    v = d.get_group("vali")
    assert list(zip(v["row"], v["col"])) == [(0, 2), (2, 0)]


def _buffered(rows, per_row, batch_mb):
    M = sp.csr_matrix(np.ones((rows, per_row)))
    opt = MatrixMarketOptions().get_default_option()
    opt.input.main = M
    opt.data.validation = {}
    opt.data.batch_mb = batch_mb
    d = MatrixMarket(opt)
    d.create()
    b = BufferedDataMatrix()
    b.initialize(d)
    return d, b


def test_buffered_single_chunk_semantics():
    d, b = _buffered(10, 6, 1024)
    b.set_group("rowwise")
    for epoch in range(2):
        sizes = list(b.fetch_batch())
        assert sizes == [60]
        start_x, next_x, indptr, keys, vals = b.get()
        assert (start_x, next_x) == (0, 10) and indptr[-1] == 60
    lind, rind, lim = b.get_indptrs()
    assert lind.shape == (10,) and rind.shape == (6,)


def test_buffered_multi_chunk_and_q24():
    # limit = max(batch_mb MB / 16, 64) = 64 -> 32 nnz per group chunk; rows of 6 nnz -> 5 rows per chunk
    d, b = _buffered(11, 6, 1e-9)
    b.set_group("rowwise")
    seen = []
    for sz in b.fetch_batch():
        start_x, next_x, indptr, keys, vals = b.get()
        beg = 0 if start_x == 0 else indptr[start_x - 1]
        assert sz == indptr[next_x - 1] - beg
        np.testing.assert_array_equal(keys[:sz], d.get_group("rowwise")["key"][beg:beg + sz])
        seen.append((start_x, next_x))
    assert seen == [(0, 5), (5, 10)]      # Q-24: the single trailing row (10) is never fed
    assert list(b.fetch_batch()) != []    # and the feeder restarts on the next epoch
