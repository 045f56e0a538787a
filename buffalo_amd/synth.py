"""Shape-matched synthetic interaction matrices (the reference's datasets are git-LFS stubs here).

Generator contract: item popularity ~ truncated Zipf, user degree ~ log-normal
scaled to the requested nnz (min 1 so every row is non-empty, max <= I/4), items drawn without
replacement per user, keys sorted ascending within a row, `indptr` = row END offsets without a
leading zero -- the layout `BufferedDataMatrix` hands to the native core
(/root/reference/buffalo/data/buffered_data.py:99-118, /root/reference/buffalo/data/fileio.hpp:359-378).
"""
import numpy as np

# BASELINE.json configs (ML-100K / ML-20M headers: /root/reference/tests/preprocess.py:19,78)
SHAPES = {
    "ml100k": (943, 1682, 80000),
    "ml20m": (138493, 27278, 20000263),
    "ml20m_i3410": (138493, 3410, 20000263),    # study shape: an item table of 1.75 MB (fits an XCD's L2); heavy users saturate it
}


class CSR:
    """Both orientations of one interaction matrix in the reference's on-disk layout."""

    def __init__(self, num_users, num_items, indptr, keys, vals):
        self.num_users, self.num_items = int(num_users), int(num_items)
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)      # [U] end offsets
        self.keys = np.ascontiguousarray(keys, dtype=np.int32)          # [nnz] item ids, sorted per row
        self.vals = np.ascontiguousarray(vals, dtype=np.float32)        # [nnz]
        self._t = None

    @property
    def nnz(self):
        return int(self.keys.shape[0])

    def rows(self):
        """Expand indptr to one row id per nnz."""
        beg = np.concatenate([[0], self.indptr[:-1]])
        return np.repeat(np.arange(self.num_users, dtype=np.int32), (self.indptr - beg))

    def transpose(self):
        """colwise group: (col,row)-sorted copy, same layout (fileio.hpp:330-341 sorts by (col,row))."""
        if self._t is None:
            rows = self.rows()
            order = np.lexsort((rows, self.keys))
            tk = rows[order].astype(np.int32)
            tv = self.vals[order]
            cnt = np.bincount(self.keys, minlength=self.num_items)
            self._t = CSR(self.num_items, self.num_users, np.cumsum(cnt, dtype=np.int64), tk, tv)
            self._t._t = self
        return self._t

    def row(self, u):
        b = 0 if u == 0 else int(self.indptr[u - 1])
        e = int(self.indptr[u])
        return self.keys[b:e], self.vals[b:e]


def _from_pairs(num_users, num_items, rows, cols, vals=None):
    key = rows.astype(np.int64) * num_items + cols.astype(np.int64)
    if vals is None:
        key = np.unique(key)
        v = None
    else:
        key, idx = np.unique(key, return_index=True)
        v = vals[idx]
    r = (key // num_items).astype(np.int64)
    c = (key % num_items).astype(np.int32)
    cnt = np.bincount(r, minlength=num_users)
    return r, c, v, cnt


def generate(num_users, num_items, nnz, seed=7, zipf_s=1.0, vals="ones", sigma=1.0):
    """ML-shaped synthetic CSR with exactly `nnz` entries (when feasible) and no empty rows."""
    rng = np.random.default_rng(seed)
    U, I = int(num_users), int(num_items)
    nnz = int(nnz)
    assert nnz >= U, "need at least one interaction per user"
    max_deg = max(1, I // 4)
    # user degrees: log-normal, rescaled to the target total
    w = rng.lognormal(mean=0.0, sigma=sigma, size=U)
    deg = np.maximum(1, np.minimum(max_deg, np.floor(w * (nnz / w.sum())).astype(np.int64)))
    for _ in range(8):  # fix the total after clamping
        diff = nnz - int(deg.sum())
        if diff == 0:
            break
        room = (max_deg - deg) if diff > 0 else (deg - 1)
        tot = int(room.sum())
        if tot == 0:
            break
        add = np.floor(room * (min(abs(diff), tot) / tot)).astype(np.int64)
        short = min(abs(diff), tot) - int(add.sum())
        if short > 0:
            cand = np.flatnonzero(room - add > 0)
            add[rng.choice(cand, size=min(short, cand.size), replace=False)] += 1
        deg = deg + add if diff > 0 else deg - add
    # item popularity: truncated Zipf over a random permutation of item ids
    p = 1.0 / np.power(np.arange(1, I + 1, dtype=np.float64), zipf_s)
    cdf = np.cumsum(p / p.sum())
    perm = rng.permutation(I).astype(np.int32)

    need = deg.copy()
    chunks = []          # accepted (user*I + item) keys, each chunk sorted & unique, chunks disjoint
    factor = 1.25
    for _ in range(200):
        users = np.flatnonzero(need > 0)
        if users.size == 0:
            break
        over = np.ceil(need[users] * factor).astype(np.int64) + 2
        r = np.repeat(users.astype(np.int64), over)
        c = perm[np.minimum(np.searchsorted(cdf, rng.random(r.shape[0])), I - 1)]
        key = r * I + c
        key = np.unique(key)
        for ch in chunks:  # drop pairs accepted in earlier rounds
            pos = np.minimum(np.searchsorted(ch, key), ch.shape[0] - 1)
            key = key[ch[pos] != key]
        # random subset per user: shuffle within user by a random secondary key
        r = key // I
        order = np.lexsort((rng.random(key.shape[0]), r))
        key = key[order]
        cnt = np.bincount(r, minlength=U)
        start = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        rank = np.arange(key.shape[0]) - np.repeat(start, cnt)
        key = np.sort(key[rank < np.repeat(need, cnt)])
        chunks.append(key)
        need = need - np.bincount(key // I, minlength=U)
        factor = min(factor * 1.5, 16.0)
    allk = np.concatenate(chunks)
    have_r, have_c = allk // I, (allk % I).astype(np.int32)
    r, c, _, cnt = _from_pairs(U, I, have_r, have_c)
    indptr = np.cumsum(cnt, dtype=np.int64)
    if vals == "ones":
        v = np.ones(c.shape[0], dtype=np.float32)
    else:  # "counts": 1 + Poisson(1), the ALS-style confidence input
        v = (1 + rng.poisson(1.0, size=c.shape[0])).astype(np.float32)
    return CSR(U, I, indptr, c, v)


def planted(num_users, num_items, d_true=8, density=0.05, seed=7, noise=0.5, popularity=0.0):
    """Small matrix with planted low-rank structure for the NDCG/MAP threshold tests that mirror
    /root/reference/tests/algo/base.py:83-97.  Returns (train CSR, held-out (user,item) pairs)."""
    rng = np.random.default_rng(seed)
    U, I = num_users, num_items
    A = rng.normal(size=(U, d_true))
    B = rng.normal(size=(I, d_true))
    S = A @ B.T + noise * rng.normal(size=(U, I))
    if popularity:   # a per-item offset skews the item popularity (head items end up in most users' rows)
        S += popularity * rng.normal(size=(1, I))
    k = max(2, int(density * I))
    top = np.argpartition(-S, k, axis=1)[:, :k]
    rows = np.repeat(np.arange(U), k)
    cols = top.reshape(-1)
    # hold one interaction per user out for validation
    held = np.zeros(rows.shape[0], dtype=bool)
    held[np.arange(U) * k + rng.integers(0, k, size=U)] = True
    r, c, _, cnt = _from_pairs(U, I, rows[~held], cols[~held])
    csr = CSR(U, I, np.cumsum(cnt, dtype=np.int64), c, np.ones(c.shape[0], dtype=np.float32))
    vali = np.stack([rows[held], cols[held]], axis=1).astype(np.int32)
    return csr, vali


def init_factors(num_users, num_items, d, seed, signed=False, vdim=None):
    """Q-18: |N(0, 1/d^2)| for BPR/ALS (bpr.py:89-94, als.py:85-86), signed for WARP (warp.py:84-89);
    `np.random.seed(seed)` only when seed != 0 (algo/base.py:33-34). Pad columns are zero."""
    if seed:
        np.random.seed(seed)
    vdim = vdim or d
    out = []
    for rows, cols in ((num_users, d), (num_items, d), (num_items, 1)):
        F = np.random.normal(scale=1.0 / (d ** 2), size=(rows, cols)).astype("float32")
        if not signed:
            F = np.abs(F)
        if cols == d and vdim > d:
            G = np.zeros((rows, vdim), dtype=np.float32)
            G[:, :d] = F
            F = G
        out.append(np.ascontiguousarray(F))
    return out
