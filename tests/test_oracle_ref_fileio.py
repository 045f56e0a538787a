"""The oracle's restatements of the data-ingestion path pinned by the REFERENCE's own compiled C++.

/root/reference/buffalo/data/fileio.hpp needs only the standard library and OpenMP, so `make -C oracle _ref` compiles it --
from where it lies, behind oracle/ref_fileio.cc's C door -- into oracle/_ref/ (git-ignored).  Where that library (or the
reference tree to build it from) exists the tests call it live; everywhere the committed outputs it produced
(tests/golden/fileio_vectors.npz, made by tests/golden/make_fileio_vectors.py) pin the same cases.

  * COO -> CSR: oracle.coo_to_csr == _sort_and_compressed_binarization, both sides, bit for bit.
  * SPPMI (CFR's context input): oracle.build_sppmi == pair lines -> psort -> _parallel_build_sppmi -> psort ->
    _chunking_into_bins -> _build_compressed_triplets, row by row as multisets, values bit for bit.  Inside a row the
    reference's order is std::unordered_set iteration order (and thread timing); the oracle and the device give (col) order.
"""
import os
import shutil
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_fileio_vectors as mk  # noqa: E402

from oracle import ref_fileio as rf  # noqa: E402

GOLDEN = np.load(mk.OUT)
live = pytest.mark.skipif(not rf.available(), reason="neither oracle/_ref/libbuffalo_fileio_ref.so nor /root/reference is here")


def _same(a, b):
    for name in ("indptr", "key"):
        assert np.array_equal(a[name], b[name]), name
    assert np.array_equal(np.asarray(a["val"], np.float32).view(np.int32), np.asarray(b["val"], np.float32).view(np.int32))


@pytest.mark.parametrize("i", range(len(mk.COO_CASES)))
def test_coo_to_csr_matches_reference_made_vectors(oracle, i):
    case = mk.COO_CASES[i]
    rows, cols, vals = mk.coo_input(case)
    _same(oracle.coo_to_csr(rows, cols, vals, case[0]), {n: GOLDEN[f"coo{i}_row_{n}"] for n in ("indptr", "key", "val")})
    _same(oracle.coo_to_csr(cols, rows, vals, case[1]), {n: GOLDEN[f"coo{i}_col_{n}"] for n in ("indptr", "key", "val")})


@pytest.mark.parametrize("i", range(len(mk.SPPMI_CASES)))
def test_sppmi_matches_reference_made_vectors(oracle, i):
    case = mk.SPPMI_CASES[i]
    indptr, items = mk.sppmi_input(case)
    o = oracle.build_sppmi(indptr, items, case[1], case[3], case[4])
    _same(rf.canonical_rows(o), {n: GOLDEN[f"sppmi{i}_{n}"] for n in ("indptr", "key", "val")})
    _same(rf.canonical_rows(o), o)      # the oracle's own order already is (row, col)


@live
def test_golden_vectors_are_what_the_reference_produces_now():
    fresh = mk.reference_vectors()
    assert sorted(fresh) == sorted(GOLDEN.files)
    for name, a in fresh.items():
        assert a.dtype == GOLDEN[name].dtype and np.array_equal(a, GOLDEN[name]), name


@live
@pytest.mark.parametrize("workers", [1, 4])
def test_coo_to_csr_live_with_more_than_one_split(oracle, workers):
    """> 4 MiB of text: the reference reads the file in 4 MiB splits on several threads (fileio.hpp:272-311)."""
    rng = np.random.default_rng(5)
    nu, ni, nnz = 30000, 8000, 450000
    rows = rng.integers(0, nu, nnz).astype(np.int32)
    cols = np.minimum((rng.pareto(1.1, nnz) * 40).astype(np.int64), ni - 1).astype(np.int32)
    vals = rng.integers(1, 6, nnz).astype(np.float32)
    _same(oracle.coo_to_csr(rows, cols, vals, nu), rf.sort_and_compressed_binarization(rows, cols, vals, nu, 1, num_workers=workers))
    _same(oracle.coo_to_csr(cols, rows, vals, ni), rf.sort_and_compressed_binarization(rows, cols, vals, ni, 2, num_workers=workers))


@live
def test_keep_order_side_of_a_stream(oracle):
    """internal_data_type 'stream' keeps the order of a user's events (stream.py:160-163, sort_key -1): the lines arrive grouped
    by user and the columns stay as they came.  oracle.coo_to_csr restates sort_key 1 and 2 only (a stream's sequences reach
    the product as (indptr, items) already); this case documents what the reference writes for -1."""
    rows = np.array([0, 0, 0, 2, 2, 5], np.int32)
    cols = np.array([4, 1, 4, 3, 0, 2], np.int32)
    vals = np.ones(6, np.float32)
    g = rf.sort_and_compressed_binarization(rows, cols, vals, 7, -1, num_workers=2)
    assert g["indptr"].tolist() == [3, 3, 5, 5, 5, 6, 6] and g["key"].tolist() == cols.tolist()


@live
@pytest.mark.parametrize("nu,ni,max_len,windows,k,workers", [(3000, 400, 60, 5, 2, 4), (20000, 3000, 40, 3, 1, 4)])
def test_sppmi_live_with_several_splits_and_workers(oracle, nu, ni, max_len, windows, k, workers):
    """6.5 MB and 19 MB of pair lines: 2 and 5 splits walked by 4 threads.  The group that straddles a split is done by the
    thread of the earlier split (fileio.hpp:182-250) and the group at end of file by nobody."""
    rng = np.random.default_rng(nu)
    lens = rng.integers(0, max_len, nu)
    indptr = np.cumsum(lens).astype(np.int64)
    items = np.minimum((rng.pareto(1.2, int(indptr[-1])) * ni / 20).astype(np.int64), ni - 2).astype(np.int32)   # top id never occurs
    ref = rf.build_sppmi(indptr, items, ni, windows, k, num_workers=workers)
    o = oracle.build_sppmi(indptr, items, ni, windows, k)
    assert ref["total_lines"] == o["total_lines"]
    _same(rf.canonical_rows(ref), o)
    last = int(items.max())
    assert (o["indptr"][last] - o["indptr"][last - 1]) == 0 and not np.any(o["key"] == last)     # the end-of-file group is absent


@live
def test_psort_restatement_is_what_sort_does(tmp_path):
    if not shutil.which("sort"):
        pytest.skip("no sort(1) here")
    rng = np.random.default_rng(0)
    lines = ["%d %d %g\n" % (rng.integers(1, 40), rng.integers(1, 1000), rng.random()) for _ in range(5000)]
    a, b = tmp_path / "a.txt", tmp_path / "b.txt"
    a.write_text("".join(lines)); b.write_text("".join(lines))
    rf._psort_first_field(str(a), 1)
    rf._psort_python(str(b), 1)
    assert a.read_text() == b.read_text()
    rf._psort_first_field(str(a), 2)
    rf._psort_python(str(b), 2)
    assert a.read_text() == b.read_text()


def test_oracle_text_parse_is_the_reference_s_parse(tmp_path):
    """oracle.parse_triples (the checker of bfh_parse_triples / bfh_text_to_csr) against the reference's own compiled fileio.hpp on the SAME text
    file: decimal values in every shape, near-ties of the float rounding, specials, CR / tab / double-space separators, with and without the
    final newline.  The reference's keep-order mode (sort_key -1) hands back (col - 1, val) per line and an indptr that is a function of the rows."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import text_cases
    from oracle import oracle as orc
    from oracle import ref_fileio as rf
    if not rf.available():
        pytest.skip("oracle/_ref/libbuffalo_fileio_ref.so is not built and /root/reference is absent")
    orc.build()
    for seed, n in ((1, 401), (2, 1000)):
        rng = np.random.default_rng(seed)
        text, n_lines = text_cases.make_text(rng, n, num_rows=60, num_cols=90, sorted_rows=True)
        rows, cols, vals = orc.parse_triples(text, n_lines)
        src = tmp_path / ("w%d.txt" % seed)
        src.write_bytes(text)
        d = tmp_path / ("out%d" % seed)
        d.mkdir()
        workers = 3
        nfiles = rf.lib().ref_sort_and_compressed_binarization(str(src).encode(), str(d).encode(), n_lines, 60, -1, workers)
        assert nfiles == workers + 1
        rec = np.dtype([("i", "<i4"), ("v", "<f4")])
        data = np.concatenate([np.fromfile(str(d / ("chunk%d.bin" % i)), dtype=rec) for i in range(workers)])
        assert data.shape[0] == n_lines
        np.testing.assert_array_equal(data["i"], cols - 1)
        np.testing.assert_array_equal(data["v"].view(np.uint32), vals.view(np.uint32))          # bit for bit, NaN payloads included
        indptr = np.fromfile(str(d / "indptr.bin"), dtype=np.int64)
        np.testing.assert_array_equal(indptr, np.cumsum(np.bincount(rows - 1, minlength=60)))
