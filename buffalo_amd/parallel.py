"""Top-k selection over factor products on the GPU -- the surface of ``buffalo.parallel``.

* ``dot_topn`` / ``quickselect`` mirror ``buffalo.parallel._core`` (/root/reference/buffalo/parallel/_core.pyx:27-56):
  same positional arguments, results written into the caller's ``out_keys`` / ``out_scores`` / ``result``.
* ``ParALS`` / ``ParBPRMF`` mirror /root/reference/buffalo/parallel/base.py:77-156 (``most_similar``,
  ``topk_recommendation``) on top of any object with ``P``, ``Q`` (``Qb``), ``opt`` and ``_idmanager``.
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


class Parallel:
    """parallel/base.py:12-75 without the N2 (hnsw) branch, which the reference never reaches either
    (`_most_similar` ignores ef_search / use_mmap, base.py:21-28)."""

    def __init__(self, algo, *argv, **kwargs):
        self.algo = algo
        self.num_workers = int(kwargs.get("num_workers", getattr(getattr(algo, "opt", None), "num_workers", 1) or 1))

    def _most_similar(self, group, indexes, Factor, topk, pool, ef_search=-1, use_mmap=True):
        dummy_bias = np.array([[]], dtype=np.float32)
        out_keys = np.zeros(shape=(len(indexes), topk), dtype=np.int32)
        out_scores = np.zeros(shape=(len(indexes), topk), dtype=np.float32)
        dot_topn(indexes, Factor, Factor, dummy_bias, out_keys, out_scores, pool, topk, self.num_workers)
        return out_keys, out_scores

    def _topk_recommendation(self, indexes, FactorP, FactorQ, topk, pool):
        dummy_bias = np.array([[]], dtype=np.float32)
        out_keys = np.zeros(shape=(len(indexes), topk), dtype=np.int32)
        out_scores = np.zeros(shape=(len(indexes), topk), dtype=np.float32)
        dot_topn(indexes, FactorP, FactorQ, dummy_bias, out_keys, out_scores, pool, topk, self.num_workers)
        return out_keys, out_scores

    def _topk_recommendation_bias(self, indexes, FactorP, FactorQ, FactorQb, topk, pool):
        out_keys = np.zeros(shape=(len(indexes), topk), dtype=np.int32)
        out_scores = np.zeros(shape=(len(indexes), topk), dtype=np.float32)
        dot_topn(indexes, FactorP, FactorQ, FactorQb, out_keys, out_scores, pool, topk, self.num_workers)
        return out_keys, out_scores


def _index_pool(algo, keys, group):
    """Algo.get_index_pool (algo/base.py:57-79 of the reference): keys -> indices, ndarray passes through."""
    if isinstance(keys, np.ndarray):
        return keys.astype(np.int32, copy=False)
    if hasattr(algo, "get_index_pool"):
        return algo.get_index_pool(keys, group=group)
    return list(keys)


class ParALS(Parallel):
    """parallel/base.py:77-131."""

    def _pool(self, pool, group):
        if pool is None:
            return np.array([], dtype=np.int32)   # empty pool means all items (base.py:91-93)
        pool = np.asarray([i for i in _index_pool(self.algo, pool, group) if i is not None], dtype=np.int32)
        if len(pool) == 0:
            raise RuntimeError("pool is empty")
        return np.ascontiguousarray(pool)

    def _queries(self, keys, group):
        indexes = _index_pool(self.algo, keys, group)
        kept = [k for k, i in zip(keys, indexes) if i is not None]
        return kept, np.ascontiguousarray([i for i in indexes if i is not None], dtype=np.int32)

    def most_similar(self, keys, topk=10, group="item", pool=None, repr=False, ef_search=-1, use_mmap=True):
        if hasattr(self.algo, "normalize"):
            self.algo.normalize(group=group)
        keys, indexes = self._queries(keys, group)
        pool = self._pool(pool, group)
        if group not in ("item", "user"):
            raise ValueError(f"Not supported group: {group}")
        F = self.algo.Q if group == "item" else self.algo.P
        topks, scores = self._most_similar(group, indexes, np.ascontiguousarray(F, dtype=np.float32), topk, pool, ef_search, use_mmap)
        if repr:
            ids = self.algo._idmanager.itemids if group == "item" else self.algo._idmanager.userids
            topks = [[ids[t] for t in tt if t != -1] for tt in topks]
        return topks, scores

    def _check_not_normalized(self):
        opt = getattr(self.algo, "opt", None)
        if opt is not None and (getattr(opt, "_nrz_P", False) or getattr(opt, "_nrz_Q", False)):
            raise RuntimeError("Cannot make topk recommendation with normalized factors")

    def topk_recommendation(self, keys, topk=10, pool=None, repr=False):
        self._check_not_normalized()
        keys, indexes = self._queries(keys, "user")
        pool = self._pool(pool, "item")
        topks, scores = self._topk_recommendation(indexes, np.ascontiguousarray(self.algo.P, dtype=np.float32),
                                                  np.ascontiguousarray(self.algo.Q, dtype=np.float32), topk, pool)
        if repr:
            topks = [[self.algo._idmanager.itemids[t] for t in tt if t != -1] for tt in topks]
        return keys, topks, scores


class ParBPRMF(ParALS):
    """parallel/base.py:134-156: the item bias joins the score."""

    def topk_recommendation(self, keys, topk=10, pool=None, repr=False):
        self._check_not_normalized()
        keys, indexes = self._queries(keys, "user")
        pool = self._pool(pool, "item")
        topks, scores = self._topk_recommendation_bias(indexes, np.ascontiguousarray(self.algo.P, dtype=np.float32),
                                                       np.ascontiguousarray(self.algo.Q, dtype=np.float32),
                                                       np.ascontiguousarray(self.algo.Qb, dtype=np.float32).reshape(-1, 1), topk, pool)
        if repr:
            topks = [[self.algo._idmanager.itemids[t] for t in tt if t != -1] for tt in topks]
        return keys, topks, scores


__all__ = ["TopK", "dot_topn", "quickselect", "Parallel", "ParALS", "ParBPRMF", "Stats", "check"]
