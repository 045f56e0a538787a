"""The stand-in loaders (tests/front_harness: MatrixMarket / Stream -> the device's COO -> CSR and SPPMI builders) against databases
built by the REFERENCE's own data package run end to end: `buffalo/data/{base,mm,stream}.py` imported unmodified, its compiled
`fileio.hpp` behind `buffalo.data.fileio` (oracle/_ref), an in-memory h5py (tests/golden/mem_h5py.py); see
tests/golden/make_data_vectors.py.  SURVEY.md section 8 f.2 cites exactly this flow (mm.py:236-279, stream.py:273-317,
data/base.py:399-451) as the step right before the hot path.

Everything the reference stores is compared: header counts, both orientations (indptr END offsets, keys, values -- bit for bit,
duplicate (row, col) entries kept apart in file order as the reference keeps them), the validation samples (same draws from
np.random, same order), the id maps, and the `sppmi` group (rows as multisets: the reference's order inside a row is
unordered_set order).  On CPU the oracle's restatements stand where the device builders are; `-m gpu` runs the same cases through
`bfh_coo_to_csr` / `bfh_sppmi_*`."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_data_vectors as mk  # noqa: E402

from oracle import ref_fileio as rf  # noqa: E402

GOLDEN = np.load(mk.OUT)
META = json.loads(str(GOLDEN["meta"]))


def _load(tmp_path, name):
    from buffalo_front.data import MatrixMarket, MatrixMarketOptions, Stream, StreamOptions
    kind, over, seed = mk.cases(str(tmp_path))[name]
    opt = (MatrixMarketOptions if kind == "mm" else StreamOptions)().get_default_option()
    mk._merge(opt, over)
    if seed is not None:
        np.random.seed(seed)
    d = (MatrixMarket if kind == "mm" else Stream)(opt)
    d.create()
    return d


def _check(d, name):
    want = {k[len(name) + 1:]: GOLDEN[k] for k in GOLDEN.files if k.startswith(name + "/")}
    header = d.get_header()
    for k in ("num_users", "num_items", "num_nnz"):
        assert header[k] == META[name]["header"][k], (k, header[k], META[name]["header"][k])
    n = header["num_nnz"]
    for g in ("rowwise", "colwise"):
        assert np.array_equal(d.get_group(g)["indptr"], want[g + "/indptr"]), (g, "indptr")
        # the reference allocates key / val before the validation samples are taken out (base.py:185-194): the tail stays zero
        assert np.array_equal(d.get_group(g)["key"], want[g + "/key"][:n]) and not want[g + "/key"][n:].any(), (g, "key")
        assert np.array_equal(d.get_group(g)["val"].view(np.int32), want[g + "/val"][:n].view(np.int32)) and not want[g + "/val"][n:].any(), g
        assert d.get_group(g)["indptr"].dtype == np.int64 and d.get_group(g)["key"].dtype == np.int32
    if "vali/row" in want and len(want["vali/row"]):
        v = d.get_group("vali")
        for a in ("row", "col", "val"):
            assert np.array_equal(v[a], want["vali/" + a]), ("vali", a, v[a][:8], want["vali/" + a][:8])
    else:
        assert not d.has_group("vali") or len(d.get_group("vali")["row"]) == 0
    assert [s.encode("utf-8") for s in d.userids] == list(want["idmap/rows"])
    assert [s.encode("utf-8") for s in d.itemids] == list(want["idmap/cols"])
    if "sppmi/key" in want:
        got = rf.canonical_rows(d.get_group("sppmi"))
        for a in ("indptr", "key"):
            assert np.array_equal(got[a], want["sppmi/" + a]), ("sppmi", a)
        assert np.array_equal(got["val"].view(np.int32), want["sppmi/val"].view(np.int32))
        assert header["sppmi_nnz"] == META[name]["sppmi_nnz"]


CASES = sorted(META)


@pytest.mark.parametrize("name", CASES)
def test_loader_over_the_oracle_builds_the_reference_database(tmp_path, oracle, monkeypatch, name):
    import buffalo_front.data as D
    monkeypatch.setattr(D, "_group", lambda nr, nc, r, c, v: oracle.coo_to_csr(r, c, v, nr, nc))
    monkeypatch.setattr(D, "_sppmi_group", lambda ip, it, ni, w, k: {n: oracle.build_sppmi(ip, it, ni, w, k)[n] for n in ("indptr", "key", "val")})
    _check(_load(tmp_path, name), name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_loader_over_the_device_builds_the_reference_database(tmp_path, name):
    _check(_load(tmp_path, name), name)


live = pytest.mark.skipif(not (rf.reference_present() and os.path.isdir(os.path.join(mk.REF, "tests", "data"))), reason="/root/reference is not here")


@live
def test_golden_databases_are_what_the_reference_builds_now(tmp_path):
    out = str(tmp_path / "fresh.npz")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_data_vectors.py"), "--out", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    fresh = np.load(out)
    assert sorted(fresh.files) == sorted(GOLDEN.files)
    for k in fresh.files:
        assert fresh[k].dtype == GOLDEN[k].dtype and np.array_equal(fresh[k], GOLDEN[k]), k


@live
def test_the_reference_s_own_data_tests_pass_in_this_setup():
    """tests/data/test_{mm,stream,prepro}.py of the reference, unmodified, over the same stand-ins: 19 tests with known answers about
    header counts, iteration order and id maps -- the setup the golden databases come from is one the reference itself accepts."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "run_reference_tests.py"), "data"], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "ran 19, failures 0, errors 0" in r.stdout
