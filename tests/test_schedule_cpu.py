"""The slice schedule of the item-major BPRMF walk (bfh_bpr_item_major_plan, host-only): whatever the queue sizes,
every slice of every queue is handed out exactly once per call, a slice never splits the slots of one positive,
and consecutive tickets land far apart in the item-sorted queue (CPU counterpart of CBPRMF's job queue,
/root/reference/include/buffalo/algo.hpp:28-69: one job per user, each taken exactly once)."""
import ctypes as C
from math import gcd

import numpy as np
import pytest


def plan(entries, num_neg, sync):
    from buffalo_amd import _lib
    L = _lib.lib()
    nq = len(entries)
    e = np.ascontiguousarray(entries, dtype=np.int64)
    slices, stride = np.zeros(nq, np.int64), np.zeros(nq, np.int64)
    slice_len, segments = C.c_int(0), C.c_int64(0)
    p64 = C.POINTER(C.c_int64)
    rc = L.bfh_bpr_item_major_plan(nq, e.ctypes.data_as(p64), num_neg, sync, C.byref(slice_len), C.byref(segments),
                                   slices.ctypes.data_as(p64), stride.ctypes.data_as(p64))
    assert rc == 0
    return slice_len.value, segments.value, slices, stride


@pytest.mark.parametrize("num_neg", [1, 2, 3, 5, 64, 65, 255])
def test_slices_hold_whole_entries_and_cover_the_queue(num_neg):
    rng = np.random.default_rng(num_neg)
    entries = rng.integers(0, 5000, size=8)
    entries[3] = 0                                         # an empty queue
    slice_len, segments, slices, stride = plan(entries, num_neg, 1 << 12)
    assert 1 <= slice_len <= 64
    if num_neg <= 64:
        assert slice_len % num_neg == 0 and slice_len + num_neg > 64
    total = int(entries.sum()) * num_neg
    assert segments == max(1, (total + (1 << 11)) >> 12)
    for x in range(8):
        triples = int(entries[x]) * num_neg
        assert slices[x] == -(-triples // slice_len)
        n = int(slices[x])
        if n == 0:
            continue
        assert 1 <= stride[x] < max(n, 2) and (n == 1 or gcd(int(stride[x]), n) == 1)
        # the ticket ranges of the segments tile [0, n) and the ticket -> slice map is a permutation
        seen = np.zeros(n, np.int64)
        prev_end = 0
        for s in range(segments):
            beg, end = n * s // segments, n * (s + 1) // segments
            assert beg == prev_end and end >= beg
            prev_end = end
            for t in range(beg, end):
                seen[(t * int(stride[x])) % n] += 1
        assert prev_end == n and np.all(seen == 1)


def test_consecutive_tickets_are_far_apart():
    """Golden-ratio order: two slices handed out back to back are at least a quarter of the queue apart, so the waves
    that run at the same time sit in different items' runs."""
    _, _, slices, stride = plan([1 << 20, 12345, 999, 64, 1, 0, 77777, 250000], 1, 1 << 23)
    for n, st in zip(slices, stride):
        n, st = int(n), int(st)
        if n >= 8:
            d = min(st, n - st)
            assert d >= n // 4, (n, st)


def test_rejects_bad_arguments():
    from buffalo_amd import _lib
    L = _lib.lib()
    one = np.ones(8, np.int64)
    p64 = C.POINTER(C.c_int64)
    sl, sg = C.c_int(0), C.c_int64(0)
    for nq, nn in ((0, 1), (9, 1), (8, 0)):
        assert L.bfh_bpr_item_major_plan(nq, one.ctypes.data_as(p64), nn, 1, C.byref(sl), C.byref(sg), one.ctypes.data_as(p64),
                                         one.ctypes.data_as(p64)) != 0


def test_shard_bounds_cost_model():
    """`shard_bounds`: contiguous ranges of equal cost, cost(row) = nnz + row_cost.  row_cost = 0 is the nnz balance the SGD walks use
    (every rank's shard timed: profiles/r04_shard_times_all_ranks.txt, slowest / mean <= 1.03); a per-row term moves the cuts towards
    equal row counts."""
    import numpy as np
    from buffalo_amd.dist import shard_bounds, shard_csr
    rng = np.random.default_rng(0)
    deg = np.concatenate([rng.integers(200, 400, 500), rng.integers(1, 5, 5000)])     # heavy head, long light tail
    indptr = np.cumsum(deg).astype(np.int64)
    for world in (2, 4, 8):
        b0 = shard_bounds(indptr, world)
        assert b0[0] == 0 and b0[-1] == len(deg) and all(x <= y for x, y in zip(b0, b0[1:]))
        nnz = [int(indptr[b - 1] - (indptr[a - 1] if a else 0)) if b > a else 0 for a, b in zip(b0, b0[1:])]
        assert max(nnz) - min(nnz) <= 2 * deg.max(), nnz
        b1 = shard_bounds(indptr, world, row_cost=1e6)                                  # rows dominate: equal row counts
        rows = [b - a for a, b in zip(b1, b1[1:])]
        assert max(rows) - min(rows) <= 2, rows
        bm = shard_bounds(indptr, world, row_cost=50.0)
        cost = [(int(indptr[b - 1] - (indptr[a - 1] if a else 0)) if b > a else 0) + 50.0 * (b - a) for a, b in zip(bm, bm[1:])]
        assert max(cost) - min(cost) <= 2 * (deg.max() + 50), cost
    u0, u1, ip, keys, off = shard_csr(indptr, np.arange(indptr[-1], dtype=np.int32), 1, 4, row_cost=50.0)
    assert ip[-1] == keys.shape[0] and keys[0] == off
