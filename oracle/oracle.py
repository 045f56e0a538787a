"""ctypes front for the CPU oracle (TEST INFRASTRUCTURE ONLY -- see buffalo_oracle.cc header).

The three classes expose the method surface of the reference's CPU Cython bindings so that
parity tests read like the reference's own tests:

* ``OracleBPRMF``  ~ ``buffalo.algo._bpr.CyBPRMF``   (/root/reference/buffalo/algo/_bpr.pyx:35-92)
* ``OracleWARP``   ~ ``buffalo.algo._warp.CyWARP``   (/root/reference/buffalo/algo/_warp.pyx)
* ``OracleALS``    ~ ``buffalo.algo._als.CyALS``     (/root/reference/buffalo/algo/_als.pyx:24-63)

That surface is exercised by the reference itself: its own ``tests/algo/test_{als,bpr,warp,eals}.py`` run unmodified with these
classes bound where its fronts import ``CyALS`` / ``CyBPRMF`` / ``CyWARP`` / ``CyEALS`` (tests/golden/run_reference_tests.py algo).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BUFFALO_ORACLE_LIB: another build of the same source (tests/test_oracle_vs_reference_sources.py loads the one without floating-point
# contraction, oracle/_ref/libbuffalo_oracle_exact.so, in a process of its own)
_LIB_PATH = os.environ.get("BUFFALO_ORACLE_LIB") or os.path.join(_HERE, "libbuffalo_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with the reference's CPU flags (oracle/Makefile)."""
    src = os.path.join(_HERE, "buffalo_oracle.cc")
    if os.environ.get("BUFFALO_ORACLE_LIB"):
        return _LIB_PATH
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libbuffalo_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    vp, i32, i64, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_double
    pf = C.POINTER(C.c_float)
    pi32 = C.POINTER(C.c_int32)
    pi64 = C.POINTER(C.c_int64)
    pu32 = C.POINTER(C.c_uint32)
    sig = {
        "orc_create": (vp, [i32]),
        "orc_destroy": (None, [vp]),
        "orc_opt_num": (None, [vp, C.c_char_p, f64]),
        "orc_opt_str": (None, [vp, C.c_char_p, C.c_char_p]),
        "orc_opt_bool": (None, [vp, C.c_char_p, i32]),
        "orc_init": (i32, [vp]),
        "orc_set_modes": (None, [vp, i32, i32, i32]),
        "orc_set_shard": (None, [vp, i64, i32]),
        "orc_trace": (None, [vp, i32]),
        "orc_trace_size": (i64, [vp]),
        "orc_trace_copy": (None, [vp, pi32]),
        "orc_sgd_initialize_model": (None, [vp, pf, i32, pf, i32, pf, i64]),
        "orc_sgd_set_cumulative_table": (None, [vp, pi64, i32]),
        "orc_sgd_launch_workers": (None, [vp]),
        "orc_sgd_add_jobs": (None, [vp, i32, i32, pi64, pi32]),
        "orc_sgd_update_parameters": (None, [vp]),
        "orc_sgd_wait_until_done": (None, [vp]),
        "orc_sgd_join": (f64, [vp]),
        "orc_sgd_compute_loss": (f64, [vp, i32, pi32, pi32, pi32]),
        "orc_bpr_apply_triples": (i32, [vp, i64, pi32, pi32, pi32, f64]),
        "orc_sgd_stats": (None, [vp, C.POINTER(C.c_longlong)]),
        "orc_sgd_state": (pf, [vp, i32, pi64]),
        "orc_bpr_exp_table": (None, [vp, pf]),
        "orc_als_initialize_model": (None, [vp, pf, i32, pf, i32]),
        "orc_als_precompute": (None, [vp, i32]),
        "orc_als_get_ff": (None, [vp, pf]),
        "orc_als_partial_update": (None, [vp, i32, i32, pi64, pi32, pf, i32, C.POINTER(f64)]),
        "orc_philox": (None, [C.c_uint32] * 6 + [pu32]),
        "orc_counter_draw": (None, [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32,
                                    C.c_uint32, pu32]),
        "orc_dot_topn": (None, [pi32, i32, pf, i32, i32, pf, i32, i32, pf, i32, pi32, pf, pi32, i32, i32, i32]),
        "orc_quickselect": (None, [pf, i32, i32, pi32, i32, i32]),
        "orc_coo_to_csr": (None, [pi32, pi32, pf, i64, i32, pi64, pi32, pf]),
        "orc_build_sppmi": (i64, [pi64, pi32, i32, i32, i32, i32, i64, pi64, pi32, pf, pi64]),
        "orc_eals_initialize_model": (None, [vp, pf, pf, pf, i32, i32]),
        "orc_eals_precompute_cache": (None, [vp, i32, pi64, pi32, i32]),
        "orc_eals_update": (i32, [vp, pi64, pi32, pf, i32]),
        "orc_eals_estimate_loss": (None, [vp, i32, pi64, pi32, pf, i32, pf]),
        "orc_eals_caches": (None, [vp, i32, pf, pi64]),
        "orc_cfr_set_embedding": (None, [vp, pf, i32, C.c_char_p]),
        "orc_cfr_precompute": (None, [vp, C.c_char_p]),
        "orc_cfr_partial_update_user": (f64, [vp, i32, i32, pi64, pi32, pf]),
        "orc_cfr_partial_update_item": (f64, [vp, i32, i32, pi64, pi32, pf, pi64, pi32, pf]),
        "orc_cfr_partial_update_context": (f64, [vp, i32, i32, pi64, pi32, pf]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _chk(a, dtype, ndim=None):
    assert isinstance(a, np.ndarray) and a.dtype == dtype and a.flags["C_CONTIGUOUS"], \
        "expected C-contiguous %s ndarray" % dtype
    if ndim is not None:
        assert a.ndim == ndim
    return a


class _Base:
    KIND = -1

    def __init__(self):
        self._h = lib().orc_create(self.KIND)
        self._keep = {}

    def __del__(self):
        try:
            if self._h:
                lib().orc_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def init(self, opt_path):
        """Reads the option JSON exactly like `Algorithm::parse_option` (lib/algo.cc:19-37):
        returns False when the file is missing or unparsable."""
        if isinstance(opt_path, bytes):
            opt_path = opt_path.decode("utf-8")
        try:
            with open(opt_path) as fin:
                opt = json.load(fin)
        except (OSError, ValueError):
            return False
        L = lib()
        for k, v in opt.items():
            kb = k.encode()
            if isinstance(v, bool):
                L.orc_opt_bool(self._h, kb, int(v))
            elif isinstance(v, (int, float)):
                L.orc_opt_num(self._h, kb, float(v))
            elif isinstance(v, str):
                L.orc_opt_str(self._h, kb, v.encode())
        return bool(L.orc_init(self._h))


class _SGDBase(_Base):
    def set_modes(self, sampler="mt19937", pos_order="unordered_set", inline=False):
        """Oracle-only switches; defaults are the reference behaviour.

        sampler: "mt19937" (reference) | "counter" (Philox draws shared with the HIP kernels)
        pos_order: "unordered_set" (reference, Q-3) | "csr"
        inline: process jobs inside add_jobs on the caller thread (deterministic lr, Q-8)
        """
        lib().orc_set_modes(self._h, {"mt19937": 0, "counter": 1}[sampler],
                            {"unordered_set": 0, "csr": 1}[pos_order], int(inline))

    def set_shard(self, nnz_offset, num_shards):
        lib().orc_set_shard(self._h, int(nnz_offset), int(num_shards))

    def state_view(self, name):
        """Zero-copy numpy view of an optimizer-state vector (used by the multi-process tests)."""
        idx = ["gradP", "gradQ", "gradQb", "momP", "momQ", "momQb", "velP", "velQ", "velQb"].index(name)
        n = C.c_int64(0)
        ptr = lib().orc_sgd_state(self._h, idx, C.byref(n))
        return np.ctypeslib.as_array(ptr, shape=(n.value,))

    def initialize_model(self, P, Q, Qb, num_total_samples):
        _chk(P, np.float32, 2), _chk(Q, np.float32, 2), _chk(Qb, np.float32, 2)
        self._keep.update(P=P, Q=Q, Qb=Qb)
        lib().orc_sgd_initialize_model(self._h, _p(P, C.c_float), P.shape[0], _p(Q, C.c_float),
                                       Q.shape[0], _p(Qb, C.c_float), int(num_total_samples))

    def set_cumulative_table(self, cum_table, size):
        _chk(cum_table, np.int64, 1)
        self._keep["cum"] = cum_table
        lib().orc_sgd_set_cumulative_table(self._h, _p(cum_table, C.c_int64), int(size))

    def launch_workers(self):
        lib().orc_sgd_launch_workers(self._h)

    def add_jobs(self, start_x, next_x, indptr, keys):
        _chk(indptr, np.int64, 1), _chk(keys, np.int32, 1)
        lib().orc_sgd_add_jobs(self._h, int(start_x), int(next_x), _p(indptr, C.c_int64),
                               _p(keys, C.c_int32))

    def update_parameters(self):
        lib().orc_sgd_update_parameters(self._h)

    def wait_until_done(self):
        lib().orc_sgd_wait_until_done(self._h)

    def join(self):
        return lib().orc_sgd_join(self._h)

    def compute_loss(self, users, positives, negatives):
        _chk(users, np.int32, 1), _chk(positives, np.int32, 1), _chk(negatives, np.int32, 1)
        return lib().orc_sgd_compute_loss(self._h, users.shape[0], _p(users, C.c_int32),
                                          _p(positives, C.c_int32), _p(negatives, C.c_int32))

    # ---- oracle-only introspection -------------------------------------------------
    def apply_triples(self, users, positives, negatives, lr):
        """BPRMF only: the SGD step of bpr.cc:119-171 applied to the given (u, pos, neg) triples in the given order."""
        n = int(users.shape[0])
        ok = lib().orc_bpr_apply_triples(self._h, n, _p(_chk(users, np.int32), C.c_int32), _p(_chk(positives, np.int32), C.c_int32),
                                         _p(_chk(negatives, np.int32), C.c_int32), float(lr))
        assert ok, "apply_triples is implemented for the BPRMF oracle only"

    def trace(self, on=True):
        lib().orc_trace(self._h, int(on))

    def get_trace(self):
        n = lib().orc_trace_size(self._h)
        out = np.zeros((n, 3), dtype=np.int32)
        if n:
            lib().orc_trace_copy(self._h, _p(out, C.c_int32))
        return out

    def stats(self):
        out = (C.c_longlong * 3)()
        lib().orc_sgd_stats(self._h, out)
        return {"samples": out[0], "scored_negatives": out[1], "updates": out[2]}

    def state(self, name):
        idx = ["gradP", "gradQ", "gradQb", "momP", "momQ", "momQb", "velP", "velQ", "velQb"].index(name)
        n = C.c_int64(0)
        ptr = lib().orc_sgd_state(self._h, idx, C.byref(n))
        if n.value == 0:
            return np.zeros(0, dtype=np.float32)
        return np.ctypeslib.as_array(ptr, shape=(n.value,)).copy()


class OracleBPRMF(_SGDBase):
    KIND = 0

    def exp_table(self):
        out = np.zeros(1000, dtype=np.float32)
        lib().orc_bpr_exp_table(self._h, _p(out, C.c_float))
        return out


class OracleWARP(_SGDBase):
    KIND = 1


class OracleALS(_Base):
    KIND = 2

    def initialize_model(self, P, Q):
        _chk(P, np.float32, 2), _chk(Q, np.float32, 2)
        self._keep.update(P=P, Q=Q)
        lib().orc_als_initialize_model(self._h, _p(P, C.c_float), P.shape[0], _p(Q, C.c_float),
                                       Q.shape[0])

    def precompute(self, axis):
        lib().orc_als_precompute(self._h, int(axis))

    def get_ff(self, d):
        out = np.zeros((d, d), dtype=np.float32)
        lib().orc_als_get_ff(self._h, _p(out, C.c_float))
        return out

    def partial_update(self, start_x, next_x, indptr, keys, vals, axis):
        _chk(indptr, np.int64, 1), _chk(keys, np.int32, 1), _chk(vals, np.float32, 1)
        out = (C.c_double * 2)()
        lib().orc_als_partial_update(self._h, int(start_x), int(next_x), _p(indptr, C.c_int64),
                                     _p(keys, C.c_int32), _p(vals, C.c_float), int(axis), out)
        return out[0], out[1]


class OracleCFR(_Base):
    """~ buffalo.algo._cfr.CyCFR (/root/reference/buffalo/algo/_cfr.pyx:25-71)."""
    KIND = 3

    def set_embedding(self, F, obj_type):
        _chk(F, np.float32, 2)
        t = obj_type.decode() if isinstance(obj_type, bytes) else obj_type
        self._keep[t] = F
        lib().orc_cfr_set_embedding(self._h, _p(F, C.c_float), F.shape[0], t.encode())

    def precompute(self, obj_type):
        t = obj_type.decode() if isinstance(obj_type, bytes) else obj_type
        lib().orc_cfr_precompute(self._h, t.encode())

    def partial_update_user(self, start_x, next_x, indptrs, keys, vals):
        _chk(indptrs, np.int64, 1), _chk(keys, np.int32, 1), _chk(vals, np.float32, 1)
        return lib().orc_cfr_partial_update_user(self._h, int(start_x), int(next_x), _p(indptrs, C.c_int64), _p(keys, C.c_int32), _p(vals, C.c_float))

    def partial_update_item(self, start_x, next_x, indptrs_u, keys_u, vals_u, indptrs_c, keys_c, vals_c):
        for a, dt in ((indptrs_u, np.int64), (keys_u, np.int32), (vals_u, np.float32), (indptrs_c, np.int64), (keys_c, np.int32), (vals_c, np.float32)):
            _chk(a, dt, 1)
        return lib().orc_cfr_partial_update_item(self._h, int(start_x), int(next_x), _p(indptrs_u, C.c_int64), _p(keys_u, C.c_int32),
                                                 _p(vals_u, C.c_float), _p(indptrs_c, C.c_int64), _p(keys_c, C.c_int32), _p(vals_c, C.c_float))

    def partial_update_context(self, start_x, next_x, indptrs, keys, vals):
        _chk(indptrs, np.int64, 1), _chk(keys, np.int32, 1), _chk(vals, np.float32, 1)
        return lib().orc_cfr_partial_update_context(self._h, int(start_x), int(next_x), _p(indptrs, C.c_int64), _p(keys, C.c_int32), _p(vals, C.c_float))


class OracleEALS(_Base):
    """~ buffalo.algo._eals.CyEALS (/root/reference/buffalo/algo/_eals.pyx:23-67)."""
    KIND = 4

    def initialize_model(self, P, Q, Cw):
        _chk(P, np.float32, 2), _chk(Q, np.float32, 2), _chk(Cw, np.float32, 1)
        self._keep.update(P=P, Q=Q, C=Cw)
        lib().orc_eals_initialize_model(self._h, _p(P, C.c_float), _p(Q, C.c_float), _p(Cw, C.c_float), P.shape[0], Q.shape[0])

    def precompute_cache(self, nnz, indptr, keys, axis):
        _chk(indptr, np.int64, 1), _chk(keys, np.int32, 1)
        lib().orc_eals_precompute_cache(self._h, int(nnz), _p(indptr, C.c_int64), _p(keys, C.c_int32), int(axis))

    def update(self, indptr, keys, vals, axis):
        _chk(indptr, np.int64, 1), _chk(keys, np.int32, 1), _chk(vals, np.float32, 1)
        return bool(lib().orc_eals_update(self._h, _p(indptr, C.c_int64), _p(keys, C.c_int32), _p(vals, C.c_float), int(axis)))

    def estimate_loss(self, nnz, indptr, keys, vals, axis):
        out = (C.c_float * 2)()
        lib().orc_eals_estimate_loss(self._h, int(nnz), _p(indptr, C.c_int64), _p(keys, C.c_int32), _p(vals, C.c_float), int(axis), out)
        return float(out[0]), float(out[1])

    def caches(self, axis, nnz):
        vhat, mp = np.empty(nnz, np.float32), np.empty(nnz, np.int64)
        lib().orc_eals_caches(self._h, int(axis), _p(vhat, C.c_float), _p(mp, C.c_int64))
        return vhat, mp


def philox4x32_10(ctr, key):
    out = (C.c_uint32 * 4)()
    lib().orc_philox(*[int(c) for c in ctr], *[int(k) for k in key], out)
    return [int(x) for x in out]


def counter_draw(seed, stream, pos_idx, slot, epoch, attempt):
    out = (C.c_uint32 * 4)()
    lib().orc_counter_draw(seed, stream, pos_idx, slot, epoch, attempt, out)
    return [int(x) for x in out]


def dot_topn(indexes, P, Q, Qb, out_keys, out_scores, pool, k, num_threads=0):
    """buffalo.parallel._core.dot_topn (_core.pyx:39-56) on the oracle; `P is Q` means "same matrix"."""
    for a, dt in ((indexes, np.int32), (P, np.float32), (Q, np.float32), (Qb, np.float32), (out_keys, np.int32),
                  (out_scores, np.float32), (pool, np.int32)):
        assert a.dtype == dt and a.flags["C_CONTIGUOUS"]
    qb_rows = Qb.shape[0] if Qb.shape[1] != 0 else 0
    same = int(P.ctypes.data == Q.ctypes.data)
    lib().orc_dot_topn(_p(indexes, C.c_int32), indexes.shape[0], _p(P, C.c_float), P.shape[0], P.shape[1],
                       _p(Q, C.c_float), Q.shape[0], Q.shape[1], _p(Qb, C.c_float), qb_rows,
                       _p(out_keys, C.c_int32), _p(out_scores, C.c_float), _p(pool, C.c_int32), pool.shape[0], int(k), same)


def quickselect(scores, result, sorted, num_threads=0):
    """buffalo.parallel._core.quickselect (_core.pyx:30-35) on the oracle."""
    assert scores.dtype == np.float32 and result.dtype == np.int32
    lib().orc_quickselect(_p(scores, C.c_float), scores.shape[0], scores.shape[1], _p(result, C.c_int32), result.shape[1], int(bool(sorted)))


def coo_to_csr(major, minor, vals, num_major, num_minor=None):
    """fileio.hpp:263-420 on the oracle: same return layout as buffalo_amd.ingest.coo_to_csr."""
    major = np.ascontiguousarray(major, dtype=np.int32)
    minor = np.ascontiguousarray(minor, dtype=np.int32)
    vals = np.ascontiguousarray(vals, dtype=np.float32)
    nnz = major.shape[0]
    indptr = np.empty(int(num_major), dtype=np.int64)
    key = np.empty(nnz, dtype=np.int32)
    val = np.empty(nnz, dtype=np.float32)
    lib().orc_coo_to_csr(_p(major, C.c_int32), _p(minor, C.c_int32), _p(vals, C.c_float), nnz, int(num_major),
                         _p(indptr, C.c_int64), _p(key, C.c_int32), _p(val, C.c_float))
    return {"indptr": indptr, "key": key, "val": val}


def parse_triples(text, total_lines):
    """fileio.hpp:280-310 on the oracle: the first `total_lines` lines of the working text file through sscanf("%d %d %f"); ids stay 1-based."""
    buf = bytes(text)
    n = int(total_lines)
    rows, cols, vals = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.float32)
    L = lib()
    L.orc_parse_triples.restype = C.c_int64
    L.orc_parse_triples.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_float)]
    found = L.orc_parse_triples(buf, len(buf), n, _p(rows, C.c_int32), _p(cols, C.c_int32), _p(vals, C.c_float))
    assert found >= n, (found, n)
    return rows, cols, vals


def build_sppmi(indptr, items, num_items, windows, k):
    """stream.py:257-267 + fileio.hpp:109-254 + stream.py:169-195 on the oracle: the SPPMI group (indptr, key, val) of a stream
    given as END-offset `indptr` [num_users] over 0-based `items`; same return layout as buffalo_amd.ingest.build_sppmi."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    items = np.ascontiguousarray(items, dtype=np.int32)
    tl = np.zeros(1, dtype=np.int64)
    out_indptr = np.zeros(int(num_items), dtype=np.int64)
    lens = np.diff(np.concatenate([[0], indptr]))
    w = int(windows)
    cap = max(1, int(2 * np.where(lens <= w + 1, lens * (lens - 1) // 2, (lens - w) * w + w * (w - 1) // 2).sum()))   # entries <= lines
    key = np.empty(cap, dtype=np.int32)
    val = np.empty(cap, dtype=np.float32)
    nnz = lib().orc_build_sppmi(_p(indptr, C.c_int64), _p(items, C.c_int32), indptr.shape[0], int(num_items), w, int(k), cap,
                                _p(out_indptr, C.c_int64), _p(key, C.c_int32), _p(val, C.c_float), _p(tl, C.c_int64))
    return {"indptr": out_indptr, "key": key[:nnz], "val": val[:nnz], "total_lines": int(tl[0])}
