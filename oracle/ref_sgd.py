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
    deps = [os.path.join(_HERE, "ref_sgd.cc"), os.path.join(_HERE, "buffalo_oracle.cc")]
    deps += [os.path.join(r, f) for r, _, fs in os.walk(os.path.join(_HERE, "stand_in_3rd")) for f in fs]
    newest = max(os.path.getmtime(p) for p in deps)
    libs = [os.path.join(_HERE, "_ref", n) for n in ("libbuffalo_bpr_on_stand_ins.so", "libbuffalo_warp_on_stand_ins.so",
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
