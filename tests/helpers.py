"""Shared drivers for the parity tests: run N epochs through the oracle and through the HIP backend
with the call sequence of `BPRMF.train` / `WARP.train` / `ALS.train`
(/root/reference/buffalo/algo/bpr.py:170-250, warp.py:187-270, als.py:115-197)."""
import json
import os
import tempfile

import numpy as np


def write_opt(opt):
    f = tempfile.NamedTemporaryFile(mode="w", suffix=".json", delete=False)
    json.dump(opt, f)
    f.close()
    return f.name


def pad(F, vdim):
    if F.shape[1] == vdim:
        return np.ascontiguousarray(F)
    G = np.zeros((F.shape[0], vdim), dtype=np.float32)
    G[:, :F.shape[1]] = F
    return G


def chunks_of(csr, n_chunks):
    """Row-aligned chunk boundaries like BufferedDataMatrix.fetch_batch (buffered_data.py:85-118)."""
    U = csr.num_users
    if n_chunks <= 1:
        return [(0, U)]
    edges = np.linspace(0, U, n_chunks + 1).astype(int)
    return [(int(a), int(b)) for a, b in zip(edges[:-1], edges[1:]) if b > a]


def chunk_arrays(csr, start_x, next_x):
    beg = 0 if start_x == 0 else int(csr.indptr[start_x - 1])
    end = int(csr.indptr[next_x - 1])
    return np.ascontiguousarray(csr.keys[beg:end]), np.ascontiguousarray(csr.vals[beg:end])


def cum_table(csr, opt):
    """bpr.py:99-111 incl. Q-4 (`int(sampling_power)`)."""
    table = np.zeros(csr.num_items, dtype=np.int64)
    if opt.get("sampling_power", 0.0) > 0.0:
        table += np.bincount(csr.keys, minlength=csr.num_items)
        table **= int(opt["sampling_power"])
        table = np.cumsum(table)
    return table


def run_oracle_sgd(orc_cls, opt, csr, P, Q, Qb, epochs, n_chunks=1, modes=None, trace=False):
    o = orc_cls()
    path = write_opt(opt)
    assert o.init(path)
    os.unlink(path)
    o.initialize_model(P, Q, Qb, csr.nnz)
    table = cum_table(csr, opt)
    o.set_cumulative_table(table, csr.num_items)
    o.set_modes(**(modes or {}))
    if trace:
        o.trace(True)
    o.launch_workers()
    for _ in range(epochs):
        for (a, b) in chunks_of(csr, n_chunks):
            keys, _ = chunk_arrays(csr, a, b)
            o.add_jobs(a, b, csr.indptr, keys)
        if not (modes or {}).get("inline"):
            _drain(o)
        o.update_parameters()
        o.wait_until_done()
    o.join()
    return o


def _drain(o):
    """Reference `wait_until_done` only waits for an empty queue; let in-flight jobs finish before
    the optimizer pass so threaded runs are comparable."""
    import time
    prev = -1
    for _ in range(2000):
        o.wait_until_done()
        cur = o.stats()["samples"]
        if cur == prev:
            return
        prev = cur
        time.sleep(0.02)


def run_hip_sgd(cls, opt, csr, P, Q, Qb, epochs, n_chunks=1, modes=None, resident=False):
    """`_prepare_train` + epochs of `_iterate` on the accelerator object (bpr.py:195-209, 170-188).
    P, Q must already be padded to vdim."""
    obj = cls()
    path = write_opt(dict(opt, accelerator=True))
    assert obj.init(path)
    os.unlink(path)
    for k, v in (modes or {}).items():
        obj.set_mode(k, v)
    obj.initialize_model(P, Q, Qb, csr.nnz)                      # init_factors (host pointers only)
    table = cum_table(csr, opt)
    obj.set_cumulative_table(table, csr.num_items)
    sizes = [int(csr.indptr[b - 1]) - (0 if a == 0 else int(csr.indptr[a - 1])) for a, b in chunks_of(csr, n_chunks)]
    obj.set_placeholder(csr.indptr, max(sizes) + 1)
    obj.initialize_model(P, Q, Qb, csr.nnz, True)
    obj.set_cumulative_table(table, csr.num_items)
    if resident:
        obj.set_resident_csr(csr.indptr, csr.keys)
    for _ in range(epochs):
        for (a, b) in chunks_of(csr, n_chunks):
            keys, _ = chunk_arrays(csr, a, b)
            obj.add_jobs(a, b, csr.indptr, None if resident else keys)
        obj.update_parameters()
        obj.wait_until_done()
    obj.synchronize(True)
    return obj


def ndcg_at_k(P, Q, train, vali, k=10, Qb=None):
    """NDCG@k of one held-out item per user, seen items masked (cf. evaluate/base.py:44-82)."""
    scores = P @ Q.T
    if Qb is not None:
        scores = scores + Qb.reshape(1, -1)
    rows = train.rows()
    scores[rows, train.keys] = -np.inf
    out = []
    for u, item in vali:
        top = np.argpartition(-scores[u], k)[:k]
        top = top[np.argsort(-scores[u][top])]
        hit = np.flatnonzero(top == item)
        out.append(1.0 / np.log2(hit[0] + 2.0) if hit.size else 0.0)
    return float(np.mean(out))


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def item_major_order(csr, nq, blocks, nn):
    """The order in which ONE wave draining every queue walks an epoch's triples (indices into the CSR-ordered
    triple list pos * nn + slot): entries stable-sorted by ((user % nq) * blocks + block) * I + item, cut into
    slices that bfh_bpr_item_major_plan hands out ticket by ticket (csrc/bpr_item_major.hpp)."""
    from test_schedule_cpu import plan   # bfh_bpr_item_major_plan through ctypes (host-only)
    n, I = csr.nnz, csr.num_items
    t = np.arange(n, dtype=np.uint64)
    blk = ((((t * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)) >> np.uint64(16)) % np.uint64(blocks)).astype(np.int64)
    key = ((csr.rows().astype(np.int64) % nq) * blocks + blk) * I + csr.keys.astype(np.int64)
    perm = np.argsort(key, kind="stable")
    queue = key[perm] // (blocks * I)
    entries = [int((queue == x).sum()) for x in range(nq)]
    slice_len, segments, slices, stride = plan(entries, nn, 1 << 40)
    assert segments == 1
    order = []
    for x in range(nq):
        trip = (perm[queue == x][:, None] * nn + np.arange(nn)[None, :]).reshape(-1)
        for ticket in range(int(slices[x])):
            sl = (ticket * int(stride[x])) % int(slices[x])
            order.append(trip[sl * slice_len:(sl + 1) * slice_len])
    return np.concatenate(order)
