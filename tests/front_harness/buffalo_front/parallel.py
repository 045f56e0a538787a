"""`Parallel` / `ParALS` / `ParBPRMF` (reference: buffalo/parallel/base.py:12-156) over `buffalo_amd.parallel.dot_topn`
(TEST INFRASTRUCTURE).  In a deployment stock buffalo's own `Par*` classes call the replaced `_core.dot_topn`
(INTEGRATION.md section 3); these stand-ins only exist so that the call traces and results of that layer can be held to the
reference's (tests/test_front_trace_cpu.py, tests/test_topk_gpu.py) without importing the reference on the GPU box."""
import numpy as np

from buffalo_amd.parallel import dot_topn


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
