"""Device-side COO -> CSR (bfh_coo_to_csr) vs the oracle's restatement of fileio.hpp:263-420: integer work,
so everything is compared bit for bit -- incl. duplicates (stable order), empty rows and both orientations."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _same(a, b):
    return all(np.array_equal(a[k], b[k]) and a[k].dtype == b[k].dtype for k in ("indptr", "key", "val"))


@pytest.mark.parametrize("n,R,C", [(1, 1, 1), (7, 3, 1000), (5000, 70, 40), (200000, 138493, 27278), (99999, 5, 3)])
def test_matches_oracle_bit_for_bit(oracle, n, R, C):
    from buffalo_amd.ingest import coo_to_csr
    rng = np.random.default_rng(n)
    r = rng.integers(0, R, n).astype(np.int32)
    c = rng.integers(0, C, n).astype(np.int32)
    v = rng.permutation(n).astype(np.float32)      # distinct: any reordering of duplicates shows
    for major, minor, nm, nn in ((r, c, R, C), (c, r, C, R)):
        assert _same(coo_to_csr(major, minor, v, nm, nn), oracle.coo_to_csr(major, minor, v, nm, nn))


def test_empty_and_invalid_inputs(oracle):
    from buffalo_amd._lib import BuffaloHipError
    from buffalo_amd.ingest import coo_to_csr
    e = np.array([], np.int32)
    g = coo_to_csr(e, e, np.array([], np.float32), 4, 9)
    assert np.array_equal(g["indptr"], np.zeros(4, np.int64)) and g["key"].shape == (0,)
    with pytest.raises(BuffaloHipError):
        coo_to_csr(np.array([0, 4], np.int32), np.array([0, 0], np.int32), np.ones(2, np.float32), 4, 9)
    with pytest.raises(BuffaloHipError):
        coo_to_csr(np.array([0, 1], np.int32), np.array([0, -1], np.int32), np.ones(2, np.float32), 4, 9)


def test_loaders_build_their_groups_on_the_device(tmp_path, oracle):
    """The MatrixMarket loader checks of tests/test_front_cpu.py, with the real device path underneath."""
    import scipy.io
    from buffalo_front.data import MatrixMarketOptions, load
    M = sp.random(40, 30, density=0.2, format="coo", random_state=3)
    M.data[:] = np.random.default_rng(0).integers(1, 5, size=M.nnz)
    scipy.io.mmwrite(str(tmp_path / "main.mtx"), M)
    opt = MatrixMarketOptions().get_default_option()
    opt.input.main = str(tmp_path / "main.mtx")
    opt.data.validation = {}
    d = load(opt)
    d.create()
    rw, cw = d.get_group("rowwise"), d.get_group("colwise")
    csr, csc = M.tocsr(), M.tocsc()
    csr.sort_indices(), csc.sort_indices()
    assert np.array_equal(rw["indptr"], csr.indptr[1:]) and np.array_equal(rw["key"], csr.indices) and np.allclose(rw["val"], csr.data)
    assert np.array_equal(cw["indptr"], csc.indptr[1:]) and np.array_equal(cw["key"], csc.indices)


def test_full_size_properties():
    """ML-20M-sized: 20,000,263 records in random order -> both orientations; sortedness, END offsets,
    and an order-independent checksum tying every (row, col, val) of the output to the input."""
    from buffalo_amd import synth
    from buffalo_amd.ingest import coo_to_csr
    csr = synth.generate(*synth.SHAPES["ml20m"], seed=7)
    U, I, nnz = csr.num_users, csr.num_items, csr.nnz
    rows = np.repeat(np.arange(U, dtype=np.int32), np.diff(np.concatenate([[0], csr.indptr])))
    perm = np.random.default_rng(1).permutation(nnz)
    r, c = rows[perm], csr.keys[perm]
    v = (perm % 1000).astype(np.float32)
    g, st = coo_to_csr(r, c, v, U, I, with_stats=True)
    assert np.array_equal(g["indptr"], csr.indptr) and np.array_equal(g["key"], csr.keys)     # the generator's CSR is (row, col)-sorted and duplicate-free
    inv = np.empty(nnz, np.int64)
    inv[perm] = np.arange(nnz)
    assert np.array_equal(g["val"], v[inv])                    # every value travelled with its record
    gt = coo_to_csr(c, r, v, I, U)
    assert gt["indptr"][-1] == nnz and np.all(np.diff(gt["indptr"]) >= 0)
    t = csr.transpose()
    assert np.array_equal(gt["indptr"], t.indptr) and np.array_equal(gt["key"], t.keys)
    assert st["samples"] == nnz and st["kernel_ms"] > 0
    print("ingest 20M records: device %.2f ms (pack + radix sort + unpack)" % st["kernel_ms"])
