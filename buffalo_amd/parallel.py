"""Top-k selection over factor products on the GPU -- the surface of ``buffalo.parallel``.

* ``dot_topn`` / ``quickselect`` mirror ``buffalo.parallel._core`` (/root/reference/buffalo/parallel/_core.pyx:27-56):
  same positional arguments, results written into the caller's ``out_keys`` / ``out_scores`` / ``result``.
* stock buffalo's own ``ParALS`` / ``ParBPRMF`` (parallel/base.py:77-156) call these two functions; they are not restated here
  (a stand-in for them lives with the test harness, tests/front_harness/buffalo_front/parallel.py).
* ``TopK.dot_topn_device`` is the resident variant: it ranks straight from the HBM buffers of a training
  handle (``CyALS`` / ``CyBPR`` / ``CyWARP``), which is what validation right after an epoch wants.

Everything runs in ``libbuffalo_hip.so`` (``bfh_topk_*``); there is no CPU fallback.
"""
import ctypes as C

import numpy as np

from ._lib import Stats, check
from .backend import _arr, _Base, _ptr


class TopK(_Base):
    _PFX = "bfh_topk_"

    def dot_topn(self, indexes, P, Q, Qb, out_keys, out_scores, pool, k):
        _arr(indexes, np.int32, 1, "indexes"), _arr(P, np.float32, 2, "P"), _arr(Q, np.float32, 2, "Q")
        _arr(Qb, np.float32, 2, "Qb"), _arr(out_keys, np.int32, 2, "out_keys"), _arr(out_scores, np.float32, 2, "out_scores")
        _arr(pool, np.int32, 1, "pool")
        k = int(k)
        if out_keys.shape != (indexes.shape[0], k) or out_scores.shape != (indexes.shape[0], k):
            raise ValueError("out_keys / out_scores must be [len(indexes), k]")
        qb_rows = Qb.shape[0] if Qb.shape[1] != 0 else 0          # _core.pyx:49
        # "same matrix" is pointer identity in the reference (_core.hpp:98); keep that through ctypes
        p_ptr = _ptr(P, C.c_float)
        q_ptr = p_ptr if P.ctypes.data == Q.ctypes.data else _ptr(Q, C.c_float)
        self._call("dot_topn", _ptr(indexes, C.c_int32), indexes.shape[0], p_ptr, P.shape[0], P.shape[1], q_ptr, Q.shape[0], Q.shape[1],
                   _ptr(Qb, C.c_float), qb_rows, _ptr(out_keys, C.c_int32), _ptr(out_scores, C.c_float), _ptr(pool, C.c_int32),
                   pool.shape[0], k)

    def dot_topn_device(self, indexes, dP, p_rows, dQ, q_rows, d, ld, dQb, same, out_keys, out_scores, pool, k):
        """dP / dQ / dQb: device addresses (ints) of row-major [rows, ld] factors, e.g. ``obj.device_buffer("Q")[0]``."""
        _arr(indexes, np.int32, 1, "indexes"), _arr(out_keys, np.int32, 2, "out_keys"), _arr(out_scores, np.float32, 2, "out_scores")
        _arr(pool, np.int32, 1, "pool")
        self._call("dot_topn_device", _ptr(indexes, C.c_int32), indexes.shape[0], C.c_void_p(dP), int(p_rows), C.c_void_p(dQ), int(q_rows),
                   int(d), int(ld), C.c_void_p(dQb or 0), int(q_rows) if dQb else 0, int(bool(same)), _ptr(out_keys, C.c_int32),
                   _ptr(out_scores, C.c_float), _ptr(pool, C.c_int32), pool.shape[0], int(k))

    def quickselect(self, scores, result, sorted=True):
        _arr(scores, np.float32, 2, "scores"), _arr(result, np.int32, 2, "result")
        if result.shape[0] != scores.shape[0]:
            raise ValueError("result must have one row per row of scores")
        self._call("quickselect", _ptr(scores, C.c_float), scores.shape[0], scores.shape[1], _ptr(result, C.c_int32), result.shape[1],
                   int(bool(sorted)))


_default = None


def _engine():
    global _default
    if _default is None:
        _default = TopK()
    return _default


def dot_topn(indexes, P, Q, Qb, out_keys, out_scores, pool, k, num_threads=0):
    """buffalo.parallel._core.dot_topn (_core.pyx:39-56); ``num_threads`` is accepted and ignored."""
    _engine().dot_topn(indexes, P, Q, Qb, out_keys, out_scores, pool, k)


def quickselect(scores, result, sorted, num_threads=0):
    """buffalo.parallel._core.quickselect (_core.pyx:27-35); rows always come back sorted."""
    _engine().quickselect(scores, result, sorted)


__all__ = ["TopK", "dot_topn", "quickselect", "Stats", "check"]
