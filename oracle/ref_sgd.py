"""The REFERENCE's own SGD classes behind ctypes (TEST INFRASTRUCTURE ONLY).

oracle/_ref/libbuffalo_{bpr,warp}_on_stand_ins.so are /root/reference/lib/algo.cc, lib/algo_impl/bpr/bpr.cc (warp/warp.cc) and
lib/misc/log.cc compiled from where they lie, unmodified, against oracle/stand_in_3rd -- stand-ins written here for the three
header-only libraries they include (Eigen, json11, spdlog) and this image lacks.  It is NOT the reference binary: how an Eigen
expression evaluates is the stand-in's reading (the same reading the oracle is written on, stand_in_3rd/README.md).  Everything
else in those files runs as the reference wrote it, which is what tests/test_oracle_vs_reference_sources.py compares the oracle with.

`RefBPRMF` / `RefWARP` have the method surface of the oracle's `OracleBPRMF` / `OracleWARP` (i.e. of CyBPRMF / CyWARP).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REFERENCE = os.environ.get("BUFFALO_REFERENCE", "/root/reference")
_libs = {}


_EXACT = bool(os.environ.get("BUFFALO_REF_SGD_EXACT"))      # the builds without floating-point contraction (oracle/Makefile: _ref_sgd_exact)


def _path(kind):
    return os.path.join(_HERE, "_ref", "libbuffalo_%s_on_stand_ins%s.so" % (kind, "_exact" if _EXACT else ""))


def oracle_exact_path():
    return os.path.join(_HERE, "_ref", "libbuffalo_oracle_exact.so")


def reference_present():
    return os.path.exists(os.path.join(_REFERENCE, "lib", "algo.cc"))


def available():
    return reference_present() or all(os.path.exists(_path(k)) for k in ("bpr", "warp"))


def build(force=False):
    if not reference_present():
        return all(os.path.exists(_path(k)) for k in ("bpr", "warp"))
    deps = [os.path.join(_HERE, "ref_sgd.cc"), os.path.join(_HERE, "ref_als.cc"), os.path.join(_HERE, "ref_cfr_eals.cc"), os.path.join(_HERE, "ref_core.cc"), os.path.join(_HERE, "buffalo_oracle.cc")]
    deps += [os.path.join(r, f) for r, _, fs in os.walk(os.path.join(_HERE, "stand_in_3rd")) for f in fs]
    newest = max(os.path.getmtime(p) for p in deps)
    libs = [os.path.join(_HERE, "_ref", n) for n in ("libbuffalo_bpr_on_stand_ins.so", "libbuffalo_warp_on_stand_ins.so", "libbuffalo_als_on_stand_ins.so", "libbuffalo_cfr_eals_on_stand_ins.so", "libbuffalo_core_on_stand_ins.so",
                                                      "libbuffalo_bpr_on_stand_ins_exact.so", "libbuffalo_warp_on_stand_ins_exact.so",
                                                      "libbuffalo_oracle_exact.so")]
    if force or any(not os.path.exists(p) or os.path.getmtime(p) < newest for p in libs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "_ref_sgd", "_ref_sgd_exact", "REFERENCE=" + _REFERENCE], stdout=subprocess.DEVNULL)
    return True


def _lib(kind):
    if kind not in _libs:
        if not build():
            raise RuntimeError("oracle/_ref/libbuffalo_%s_on_stand_ins.so is not built and /root/reference is absent" % kind)
        L = C.CDLL(_path(kind))
        p = "refsgd_%s_" % kind
        vp, i32, f64 = C.c_void_p, C.c_int, C.c_double
        pf, pi32, pi64 = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        for name, (res, args) in {
            "create": (vp, []), "destroy": (None, [vp]), "init": (i32, [vp, C.c_char_p]),
            "initialize_model": (None, [vp, pf, i32, pf, i32, pf, C.c_longlong]), "set_cumulative_table": (None, [vp, pi64, i32]),
            "launch_workers": (None, [vp]), "add_jobs": (None, [vp, i32, i32, pi64, pi32]), "wait_until_done": (None, [vp]),
            "update_parameters": (None, [vp]), "join": (f64, [vp]), "compute_loss": (f64, [vp, i32, pi32, pi32, pi32]),
            "queue_size": (i32, [vp]),
        }.items():
            fn = getattr(L, p + name)
            fn.restype, fn.argtypes = res, args
        _libs[kind] = L
    return _libs[kind]


def _ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


class _RefSGD:
    KIND = None

    def __init__(self):
        self._L = _lib(self.KIND)
        self._h = self._call("create")
        self._keep = {}

    def _call(self, name, *args):
        return getattr(self._L, "refsgd_%s_%s" % (self.KIND, name))(*args)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._call("destroy", self._h)
                self._h = None
        except Exception:
            pass

    def init(self, opt_path):
        if isinstance(opt_path, str):
            opt_path = opt_path.encode("utf-8")
        return bool(self._call("init", self._h, opt_path))

    def initialize_model(self, P, Q, Qb, num_total_samples):
        for a in (P, Q, Qb):
            assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
        self._keep.update(P=P, Q=Q, Qb=Qb)
        self._call("initialize_model", self._h, _ptr(P, C.c_float), P.shape[0], _ptr(Q, C.c_float), Q.shape[0], _ptr(Qb, C.c_float), int(num_total_samples))

    def set_cumulative_table(self, table, size):
        assert table.dtype == np.int64
        self._keep["cum"] = table
        self._call("set_cumulative_table", self._h, _ptr(table, C.c_int64), int(size))

    def launch_workers(self):
        self._call("launch_workers", self._h)

    def add_jobs(self, start_x, next_x, indptr, keys):
        assert indptr.dtype == np.int64 and keys.dtype == np.int32
        self._call("add_jobs", self._h, int(start_x), int(next_x), _ptr(indptr, C.c_int64), _ptr(keys, C.c_int32))

    def wait_until_done(self):
        self._call("wait_until_done", self._h)

    def queue_size(self):
        return self._call("queue_size", self._h)

    def update_parameters(self):
        self._call("update_parameters", self._h)

    def join(self):
        return self._call("join", self._h)

    def compute_loss(self, users, positives, negatives):
        return self._call("compute_loss", self._h, len(users), _ptr(users, C.c_int32), _ptr(positives, C.c_int32), _ptr(negatives, C.c_int32))


class RefBPRMF(_RefSGD):
    KIND = "bpr"


class RefWARP(_RefSGD):
    KIND = "warp"


class RefALS:
    """~ OracleALS / CyALS (buffalo/algo/_als.pyx:24-63) over the reference's als.cc on the stand-ins (no contraction-free twin: its dense
    products and solves are the stand-in's own loops, compared with a tolerance)."""

    def __init__(self):
        if not build():
            raise RuntimeError("oracle/_ref/libbuffalo_als_on_stand_ins.so is not built and /root/reference is absent")
        L = self._L = C.CDLL(os.path.join(_HERE, "_ref", "libbuffalo_als_on_stand_ins.so"))
        vp, i32 = C.c_void_p, C.c_int
        pf, pi32, pi64, pd = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double)
        for name, (res, args) in {"create": (vp, []), "destroy": (None, [vp]), "init": (i32, [vp, C.c_char_p]),
                                  "initialize_model": (None, [vp, pf, i32, pf, i32]), "precompute": (None, [vp, i32]),
                                  "partial_update": (None, [vp, i32, i32, pi64, pi32, pf, i32, pd]), "get_ff": (None, [vp, pf, i32])}.items():
            fn = getattr(L, "refals_" + name)
            fn.restype, fn.argtypes = res, args
        self._h = L.refals_create()
        self._keep = {}

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.refals_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def init(self, opt_path):
        if isinstance(opt_path, str):
            opt_path = opt_path.encode("utf-8")
        return bool(self._L.refals_init(self._h, opt_path))

    def initialize_model(self, P, Q):
        assert P.dtype == np.float32 and Q.dtype == np.float32 and P.flags["C_CONTIGUOUS"] and Q.flags["C_CONTIGUOUS"]
        self._keep.update(P=P, Q=Q)
        self._L.refals_initialize_model(self._h, _ptr(P, C.c_float), P.shape[0], _ptr(Q, C.c_float), Q.shape[0])

    def precompute(self, axis):
        self._L.refals_precompute(self._h, int(axis))

    def get_ff(self, d):
        out = np.empty((d, d), np.float32)
        self._L.refals_get_ff(self._h, _ptr(out, C.c_float), int(d))
        return out

    def partial_update(self, start_x, next_x, indptr, keys, vals, axis):
        assert indptr.dtype == np.int64 and keys.dtype == np.int32 and vals.dtype == np.float32
        out = (C.c_double * 2)()
        self._L.refals_partial_update(self._h, int(start_x), int(next_x), _ptr(indptr, C.c_int64), _ptr(keys, C.c_int32), _ptr(vals, C.c_float), int(axis), out)
        return float(out[0]), float(out[1])


def _cfr_eals_lib():
    if "cfr_eals" not in _libs:
        if not build():
            raise RuntimeError("oracle/_ref/libbuffalo_cfr_eals_on_stand_ins.so is not built and /root/reference is absent")
        L = C.CDLL(os.path.join(_HERE, "_ref", "libbuffalo_cfr_eals_on_stand_ins.so"))
        vp, i32, f64 = C.c_void_p, C.c_int, C.c_double
        pf, pi32, pi64 = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        for name, (res, args) in {
            "refcfr_create": (vp, []), "refcfr_destroy": (None, [vp]), "refcfr_init": (i32, [vp, C.c_char_p]),
            "refcfr_set_embedding": (None, [vp, pf, i32, C.c_char_p]), "refcfr_precompute": (None, [vp, C.c_char_p]),
            "refcfr_partial_update_user": (f64, [vp, i32, i32, pi64, pi32, pf]),
            "refcfr_partial_update_item": (f64, [vp, i32, i32, pi64, pi32, pf, pi64, pi32, pf]),
            "refcfr_partial_update_context": (f64, [vp, i32, i32, pi64, pi32, pf]),
            "refeals_create": (vp, []), "refeals_destroy": (None, [vp]), "refeals_init": (i32, [vp, C.c_char_p]),
            "refeals_initialize_model": (None, [vp, pf, pf, pf, i32, i32]), "refeals_precompute_cache": (None, [vp, i32, pi64, pi32, i32]),
            "refeals_update": (i32, [vp, pi64, pi32, pf, i32]), "refeals_estimate_loss": (None, [vp, i32, pi64, pi32, pf, i32, pf]),
        }.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _libs["cfr_eals"] = L
    return _libs["cfr_eals"]


class RefCFR:
    """~ OracleCFR / CyCFR (buffalo/algo/_cfr.pyx:25-71) over the reference's cfr.cc on the stand-ins."""

    def __init__(self):
        self._L = _cfr_eals_lib()
        self._h = self._L.refcfr_create()
        self._keep = {}

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.refcfr_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @staticmethod
    def _b(x):
        return x if isinstance(x, bytes) else str(x).encode("utf-8")

    def init(self, opt_path):
        return bool(self._L.refcfr_init(self._h, self._b(opt_path)))

    def set_embedding(self, F, obj_type):
        assert F.dtype == np.float32 and F.flags["C_CONTIGUOUS"]
        self._keep[self._b(obj_type)] = F
        self._L.refcfr_set_embedding(self._h, _ptr(F, C.c_float), F.shape[0], self._b(obj_type))

    def precompute(self, obj_type):
        self._L.refcfr_precompute(self._h, self._b(obj_type))

    def partial_update_user(self, start_x, next_x, indptrs, keys, vals):
        return self._L.refcfr_partial_update_user(self._h, int(start_x), int(next_x), _ptr(indptrs, C.c_int64), _ptr(keys, C.c_int32), _ptr(vals, C.c_float))

    def partial_update_item(self, start_x, next_x, indptrs_u, keys_u, vals_u, indptrs_c, keys_c, vals_c):
        return self._L.refcfr_partial_update_item(self._h, int(start_x), int(next_x), _ptr(indptrs_u, C.c_int64), _ptr(keys_u, C.c_int32),
                                                  _ptr(vals_u, C.c_float), _ptr(indptrs_c, C.c_int64), _ptr(keys_c, C.c_int32), _ptr(vals_c, C.c_float))

    def partial_update_context(self, start_x, next_x, indptrs, keys, vals):
        return self._L.refcfr_partial_update_context(self._h, int(start_x), int(next_x), _ptr(indptrs, C.c_int64), _ptr(keys, C.c_int32), _ptr(vals, C.c_float))


class RefEALS:
    """~ OracleEALS / CyEALS (buffalo/algo/_eals.pyx:23-67) over the reference's eals.cc / eals.hpp on the stand-ins (+ a written-out ssyrk)."""

    def __init__(self):
        self._L = _cfr_eals_lib()
        self._h = self._L.refeals_create()
        self._keep = {}

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.refeals_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def init(self, opt_path):
        return bool(self._L.refeals_init(self._h, opt_path if isinstance(opt_path, bytes) else str(opt_path).encode("utf-8")))

    def initialize_model(self, P, Q, Cw):
        self._keep.update(P=P, Q=Q, C=Cw)
        self._L.refeals_initialize_model(self._h, _ptr(P, C.c_float), _ptr(Q, C.c_float), _ptr(Cw, C.c_float), P.shape[0], Q.shape[0])

    def precompute_cache(self, nnz, indptr, keys, axis):
        self._L.refeals_precompute_cache(self._h, int(nnz), _ptr(indptr, C.c_int64), _ptr(keys, C.c_int32), int(axis))

    def update(self, indptr, keys, vals, axis):
        return bool(self._L.refeals_update(self._h, _ptr(indptr, C.c_int64), _ptr(keys, C.c_int32), _ptr(vals, C.c_float), int(axis)))

    def estimate_loss(self, nnz, indptr, keys, vals, axis):
        out = (C.c_float * 2)()
        self._L.refeals_estimate_loss(self._h, int(nnz), _ptr(indptr, C.c_int64), _ptr(keys, C.c_int32), _ptr(vals, C.c_float), int(axis), out)
        return float(out[0]), float(out[1])


def _core_lib():
    if "core" not in _libs:
        if not build():
            raise RuntimeError("oracle/_ref/libbuffalo_core_on_stand_ins.so is not built and /root/reference is absent")
        L = C.CDLL(os.path.join(_HERE, "_ref", "libbuffalo_core_on_stand_ins.so"))
        i32 = C.c_int
        pf, pi32 = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.refcore_quickselect.restype, L.refcore_quickselect.argtypes = None, [pf, i32, i32, pi32, i32, i32, i32]
        L.refcore_dot_topn.restype, L.refcore_dot_topn.argtypes = None, [pi32, i32, pf, i32, i32, pf, i32, i32, pf, i32, pi32, pf, pi32, i32, i32, i32]
        _libs["core"] = L
    return _libs["core"]


def dot_topn(indexes, P, Q, Qb, out_keys, out_scores, pool, k, num_threads=1):
    """buffalo.parallel._core.dot_topn (_core.pyx:38-58) over the reference's _core.hpp on the stand-ins; same arguments as oracle.dot_topn."""
    qb_rows = Qb.shape[0] if Qb.ndim == 2 and Qb.shape[1] != 0 else 0
    _core_lib().refcore_dot_topn(_ptr(indexes, C.c_int32), indexes.shape[0], _ptr(P, C.c_float), P.shape[0], P.shape[1], _ptr(Q, C.c_float), Q.shape[0],
                                 Q.shape[1], _ptr(Qb, C.c_float), qb_rows, _ptr(out_keys, C.c_int32), _ptr(out_scores, C.c_float),
                                 _ptr(pool, C.c_int32), pool.shape[0], int(k), int(num_threads))


def quickselect(scores, result, sorted, num_threads=1):
    """buffalo.parallel._core.quickselect (_core.pyx:30-35) over the reference's _core.hpp on the stand-ins."""
    _core_lib().refcore_quickselect(_ptr(scores, C.c_float), scores.shape[0], scores.shape[1], _ptr(result, C.c_int32), result.shape[1], int(bool(sorted)),
                                    int(num_threads))
