"""Multi-GPU data parallelism, host side of the product path: how the rows are cut over the ranks, and the ALS epoch loop on the
library's own communicator.  One process per GPU; everything that travels between GPUs travels INSIDE the library
(`obj.set_comm(Comm(...))`: csrc/sgd_base.hip exchange_*, csrc/als_kernels.hpp publish_rows, csrc/comm.hip = RCCL over xGMI).

The reference has no multi-device code at all; the scheme is new:
* users (P rows, their CSR rows, their optimizer state) are sharded into contiguous, nnz-balanced ranges -- one shard per rank; a
  triple touches one user row and two item rows, so user shards never interact;
* item factors Q (+Qb) are replicated; what a rank changed in them since the state every rank agrees on is summed over the ranks
  with ONE all-reduce per exchange point (sgd: the model deltas, weighted between sum and mean per row; adam / adagrad / WARP:
  the gradient deltas, after which every rank takes the identical optimizer step);
* ALS: the rows being solved are sharded, both factor matrices replicated, the solved row blocks are published after each
  half-epoch.

The torch.distributed restatement of the exchange protocol that the gloo tests run on CPU lives in tests/dist_harness.py
(test infrastructure; one protocol, ONE product implementation -- the library's).
"""
import numpy as np


def shard_bounds(indptr, world_size, row_cost=0.0):
    """Contiguous user ranges of ~equal COST, cost(row) = nnz(row) + row_cost (degree distributions are heavy-tailed, so equal
    row counts would not balance; `row_cost` -- what a row costs beyond its entries, in entries -- is 0 for the SGD walks, whose
    per-shard time follows the nnz to a few percent: profiles/r04_shard_times_all_ranks.txt).  Returns world_size+1 row boundaries."""
    indptr = np.asarray(indptr, dtype=np.int64)
    U = indptr.shape[0]
    nnz = int(indptr[-1]) if U else 0
    cum = indptr.astype(np.float64) + row_cost * np.arange(1, U + 1, dtype=np.float64) if row_cost else indptr
    total = float(cum[-1]) if U else 0.0
    bounds = [0]
    for r in range(1, world_size):
        target = (nnz * r // world_size) if not row_cost else total * r / world_size
        b = int(np.searchsorted(cum, target, side="left")) + 1  # first row whose end offset >= target
        bounds.append(min(max(b, bounds[-1]), U))
    bounds.append(U)
    return bounds


def shard_csr(indptr, keys, rank, world_size, row_cost=0.0):
    """Rank-local CSR: rows [u0,u1) with indptr rebased to start at zero.
    Returns (u0, u1, local_indptr, local_keys, nnz_offset)."""
    b = shard_bounds(indptr, world_size, row_cost)
    u0, u1 = b[rank], b[rank + 1]
    beg = 0 if u0 == 0 else int(indptr[u0 - 1])
    end = beg if u1 == u0 else int(indptr[u1 - 1])
    local_indptr = np.ascontiguousarray(np.asarray(indptr[u0:u1], dtype=np.int64) - beg)
    return u0, u1, local_indptr, np.ascontiguousarray(keys[beg:end]), beg


class CommDataParallelALS:
    """DataParallelALS on the library's own communicator: `obj` is a CyALS with both orientations resident and
    `obj.set_comm(comm)` done; the solved row blocks travel with `bfh_als_publish_rows` (a group of ncclBroadcast)."""

    def __init__(self, obj, comm, indptrs, num_users, num_items):
        self.obj, self.comm, self.indptr, self.rows = obj, comm, indptrs, (num_users, num_items)
        self.bounds = [shard_bounds(ip, comm.world) for ip in indptrs]
        obj.set_mode("als_writeback", 0)

    def half_epoch(self, axis):
        self.obj.precompute(axis)
        b = self.bounds[axis]
        nume, deno = self.obj.partial_update(b[self.comm.rank], b[self.comm.rank + 1], self.indptr[axis], None, None, axis)
        self.obj.publish_rows(axis, b)
        return tuple(self.comm.all_reduce([nume, deno])) if self.comm.world > 1 else (nume, deno)

    def epoch(self):
        n0, d0 = self.half_epoch(0)
        n1, d1 = self.half_epoch(1)
        return n0 + n1, d0 + d1


