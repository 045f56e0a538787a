"""Error behaviour at the boundary (SURVEY.md section 8(b) "Errors"): the reference's `init` returns False on
an unreadable / invalid option file and every other failure surfaces as an exception through Cython
(`except +`, cuda/_bpr.pyx:14-23).  Here: `init` -> False, everything else -> BuffaloHipError carrying
bfh_last_error(); nothing may crash the process or leave the handle unusable."""
import numpy as np
import pytest

import helpers as H
from conftest import als_opt, bpr_opt, tiny_csr, warp_opt

pytestmark = pytest.mark.gpu


def _bpr(opt=None):
    from buffalo_amd.backend import CyBPR
    obj = CyBPR()
    if opt is not None:
        assert obj.init(H.write_opt(dict(opt, accelerator=True)))
    return obj


def test_init_returns_false_on_bad_option_files(tmp_path):
    from buffalo_amd.backend import CyALS, CyBPR, CyWARP
    missing = str(tmp_path / "nope.json")
    garbage = tmp_path / "garbage.json"
    garbage.write_text("{ this is not json")
    incomplete = tmp_path / "incomplete.json"
    incomplete.write_text('{"d": 8}')
    from buffalo_amd._lib import BuffaloHipError
    for cls in (CyBPR, CyWARP, CyALS):
        for path in (missing, str(garbage)):
            assert cls().init(path) is False
        # json11 hands the reference 0 / "" for a missing key and training then runs on nonsense; here a missing
        # required option is an error that names the key
        with pytest.raises(BuffaloHipError, match="missing"):
            cls().init(str(incomplete))
    obj = CyBPR()
    assert obj.init(missing) is False
    assert obj.init(H.write_opt(bpr_opt(d=8, accelerator=True))) is True      # the handle survives a failed init


def test_unsupported_options_fail_loudly():
    from buffalo_amd._lib import BuffaloHipError
    from buffalo_amd.backend import CyALS, CyBPR
    for bad in (dict(optimizer="rmsprop"), dict(d=0), dict(d=4096)):
        obj = CyBPR()
        try:
            ok = obj.init(H.write_opt(bpr_opt(accelerator=True, **dict(dict(d=8), **bad))))
        except BuffaloHipError:
            ok = False
        assert ok is False, bad
    obj = CyALS()
    try:     # Eigen's Krylov solvers (algo.cc:84-126) are outside the scope table: refused at init, not silently replaced
        ok = obj.init(H.write_opt(als_opt(d=8, optimizer="eigen_cg", accelerator=True)))
    except BuffaloHipError:
        ok = False
    assert ok is False


def test_call_order_and_range_errors_keep_the_handle_usable():
    from buffalo_amd._lib import BuffaloHipError
    csr = tiny_csr(U=30, I=20, density=0.2, seed=1)
    d, vdim = 8, 32
    rng = np.random.default_rng(0)
    P = H.pad(rng.normal(size=(30, d)).astype(np.float32), vdim)
    Q = H.pad(rng.normal(size=(20, d)).astype(np.float32), vdim)
    Qb = np.zeros((20, 1), np.float32)
    obj = _bpr(bpr_opt(d=d))
    with pytest.raises(BuffaloHipError):                      # add_jobs before the model is on the device
        obj.add_jobs(0, 30, csr.indptr, csr.keys)
    obj.initialize_model(P, Q, Qb, csr.nnz, True)
    obj.set_cumulative_table(np.zeros(20, np.int64), 20)
    with pytest.raises(BuffaloHipError):                      # no placeholder / resident CSR yet
        obj.add_jobs(0, 30, csr.indptr, csr.keys)
    obj.set_placeholder(csr.indptr, csr.nnz)
    for a, b in ((-1, 5), (5, 3), (0, 31)):
        with pytest.raises(BuffaloHipError):
            obj.add_jobs(a, b, csr.indptr, csr.keys)
    with pytest.raises(BuffaloHipError):
        obj.set_mode("no_such_knob", 1)
    before = Q.copy()
    loss, n = obj.add_jobs(0, 30, csr.indptr, csr.keys)      # ... and the handle still trains
    obj.update_parameters()
    assert n == 0 or n == csr.nnz or n >= 0
    assert not np.array_equal(Q, before) and np.isfinite(Q).all()
    with pytest.raises(ValueError):                           # Cython-style typed-buffer check, before the C ABI
        obj.add_jobs(0, 30, csr.indptr.astype(np.int32), csr.keys)


def test_als_and_warp_argument_checks():
    from buffalo_amd._lib import BuffaloHipError
    from buffalo_amd.backend import CyALS, CyWARP
    csr = tiny_csr(U=12, I=9, density=0.3, seed=2, counts=True)
    t = csr.transpose()
    P = H.pad(np.abs(np.random.default_rng(1).normal(size=(12, 8))).astype(np.float32), 32)
    Q = H.pad(np.abs(np.random.default_rng(2).normal(size=(9, 8))).astype(np.float32), 32)
    als = CyALS()
    assert als.init(H.write_opt(als_opt(d=8, accelerator=True)))
    with pytest.raises(BuffaloHipError):
        als.precompute(0)                                      # before initialize_model
    als.initialize_model(P, Q)
    with pytest.raises(BuffaloHipError):
        als.partial_update(0, 12, csr.indptr, csr.keys, csr.vals, 0)   # before set_placeholder
    als.set_placeholder(csr.indptr, t.indptr, csr.nnz)
    als.precompute(0)
    for axis in (2, -1):
        with pytest.raises(BuffaloHipError):
            als.partial_update(0, 12, csr.indptr, csr.keys, csr.vals, axis)
    with pytest.raises(BuffaloHipError):
        als.partial_update(0, 13, csr.indptr, csr.keys, csr.vals, 0)
    assert als.partial_update(3, 3, csr.indptr, csr.keys[:0], csr.vals[:0], 0) == (0.0, 0.0)      # als.cc:219-222: empty range is a no-op
    nume, deno = als.partial_update(0, 12, csr.indptr, csr.keys, csr.vals, 0)
    assert np.isfinite(P).all() and np.isfinite(nume)
    warp = CyWARP()
    assert warp.init(H.write_opt(warp_opt(d=8, accelerator=True)))
    with pytest.raises(BuffaloHipError):
        warp.add_jobs(0, 12, csr.indptr, csr.keys)
    with pytest.raises(ValueError):
        warp.initialize_model(P[:, :8], Q, np.zeros((9, 1), np.float32), csr.nnz, True)   # not padded to vdim


def test_topk_argument_checks():
    from buffalo_amd import parallel as par
    from buffalo_amd._lib import BuffaloHipError
    import topk_cases as tc
    P = tc.integer_factors(6, 4, seed=1)
    Q = tc.integer_factors(9, 4, seed=2)
    eng = par.TopK()
    ok, os_ = np.empty((2, 3), np.int32), np.empty((2, 3), np.float32)
    with pytest.raises(BuffaloHipError):
        eng.dot_topn(np.array([0, 6], np.int32), P, Q, tc.NO_BIAS, ok, os_, tc.EMPTY_POOL, 3)      # query outside P
    with pytest.raises(BuffaloHipError):
        eng.dot_topn(np.array([0, 1], np.int32), P, tc.integer_factors(9, 5, seed=3), tc.NO_BIAS, ok, os_, tc.EMPTY_POOL, 3)
    with pytest.raises(ValueError):
        eng.dot_topn(np.array([0, 1], np.int32), P, Q, tc.NO_BIAS, np.empty((2, 4), np.int32), os_, tc.EMPTY_POOL, 3)
    big = np.empty((2, 20000), np.int32), np.empty((2, 20000), np.float32)
    with pytest.raises(BuffaloHipError):
        eng.dot_topn(np.array([0, 1], np.int32), P, Q, tc.NO_BIAS, big[0], big[1], tc.EMPTY_POOL, 20000)   # k > 16384
    eng.dot_topn(np.array([0, 1], np.int32), P, Q, tc.NO_BIAS, ok, os_, tc.EMPTY_POOL, 3)               # still usable
    assert (ok >= -1).all()
    empty = np.empty((0, 3), np.int32), np.empty((0, 3), np.float32)
    eng.dot_topn(np.array([], np.int32), P, Q, tc.NO_BIAS, empty[0], empty[1], tc.EMPTY_POOL, 3)        # zero queries: no-op


@pytest.mark.parametrize("pin", [0, 1])
def test_freed_caller_arrays_are_an_exception_not_a_fault(pin):
    """The backend stores raw host pointers and writes the model back into them (bpr.cu:334-336).  With `lazy_sync` = 1 the copy is
    owed until flush / destroy; if the caller's arrays are gone by then (here: unmapped -- large numpy arrays are private mmaps) the
    library has to say so (BFH_ERR_INVALID from the mincore probe in front of every device -> caller copy), not fault.  `pin` = 1:
    the same with the arrays registered (`pin_host`, opt-in since round 5; the default copies back through the library's own pinned ring)."""
    import gc
    import mmap
    from buffalo_amd._lib import BuffaloHipError
    d, U, I = 128, 6000, 5000
    csr = tiny_csr(U=U, I=I, density=0.002, seed=3)
    obj = _bpr(bpr_opt(d=d, lr=0.01, min_lr=0.01, num_iters=2))
    obj.set_mode("lazy_sync", 1)
    obj.set_mode("pin_host", pin)
    obj.sync_every_epoch = True
    # page-aligned private mappings the test can unmap for sure (numpy would hand a freed array back to malloc, which may keep the pages)
    maps = [mmap.mmap(-1, n * 4) for n in (U * d, I * d, I)]
    P, Q, Qb = (np.frombuffer(m, dtype=np.float32).reshape(shape) for m, shape in zip(maps, ((U, d), (I, d), (I, 1))))
    rng = np.random.default_rng(1)
    P[:], Q[:], Qb[:] = rng.normal(scale=0.1, size=P.shape), rng.normal(scale=0.1, size=Q.shape), 0
    obj.initialize_model(P, Q, Qb, csr.nnz, True)
    obj.set_cumulative_table(np.zeros(I, np.int64), I)
    obj.set_placeholder(csr.indptr, csr.nnz + 1)
    obj.add_jobs(0, U, csr.indptr, csr.keys)
    obj.update_parameters()                     # lazy: the arrays are now stale, the copy is owed
    P0 = P.copy()
    obj.flush_host()                            # while the arrays live: the copy arrives
    assert not np.array_equal(P, P0) and np.isfinite(P).all()
    obj.add_jobs(0, U, csr.indptr, csr.keys)
    obj.update_parameters()
    obj._keep.clear()                           # what a C caller can do: free the arrays while the handle still owes them a copy
    del P, Q, Qb, P0
    gc.collect()
    for m in maps:
        m.close()                               # munmap
    with pytest.raises(BuffaloHipError, match="no longer mapped"):
        obj.flush_host()
    # the handle is still usable with new arrays (initialize_model drops the debt to the old ones: it cannot be paid)
    P2, Q2, Qb2 = rng.normal(scale=0.1, size=(U, d)).astype(np.float32), rng.normal(scale=0.1, size=(I, d)).astype(np.float32), np.zeros((I, 1), np.float32)
    try:
        obj.initialize_model(P2, Q2, Qb2, csr.nnz, True)
    except BuffaloHipError as e:                # the owed copy is attempted first and refused: the second call starts clean
        assert "no longer mapped" in str(e)
        obj.initialize_model(P2, Q2, Qb2, csr.nnz, True)
    obj.add_jobs(0, U, csr.indptr, csr.keys)
    obj.update_parameters()
    obj.flush_host()
    assert np.isfinite(P2).all()
    del obj
