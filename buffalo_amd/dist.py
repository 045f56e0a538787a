"""Multi-GPU data parallelism (BPRMF / WARP: DataParallelSGD, ALS: DataParallelALS): one process per GPU,
`torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no multi-device code at all; the scheme below is new:

* users (P rows, their CSR rows, their optimizer state) are sharded into contiguous, nnz-balanced
  ranges -- one shard per rank; a triple touches one user row and two item rows, so user shards
  never interact;
* item factors Q (+Qb) are replicated.  What a rank changed in the replicated tensors during one
  minibatch (= one `add_jobs` call over its shard) is exchanged as a *delta* with ONE all-reduce:
      T_new = T_sync + sum_r (T_r - T_sync)
  - optimizer "sgd" (Hogwild): T = {Q, Qb}  -> local-SGD with summed updates;
  - adam / adagrad / WARP:     T = {gradQ, gradQb, countQ}; the optimizer step is then computed
    redundantly and identically on every rank, which is *exactly* the single-GPU result up to fp32
    summation order (the delta form keeps the never-re-zeroed gradient residue of Q-6 from being
    counted world_size times).
  Q is 14 MB at ML-20M/d=128 and 1 GB at the 10M x 1M WARP config: one collective per minibatch
  keeps the ring all-reduce (per-link bound on xGMI) off the critical path.
"""
import numpy as np


def shard_bounds(indptr, world_size):
    """Contiguous user ranges with ~equal nnz (degree distributions are heavy-tailed, so equal
    row counts would not balance).  Returns world_size+1 row boundaries."""
    indptr = np.asarray(indptr, dtype=np.int64)
    U = indptr.shape[0]
    nnz = int(indptr[-1]) if U else 0
    bounds = [0]
    for r in range(1, world_size):
        target = nnz * r // world_size
        b = int(np.searchsorted(indptr, target, side="left")) + 1  # first row whose end offset >= target
        bounds.append(min(max(b, bounds[-1]), U))
    bounds.append(U)
    return bounds


def shard_csr(indptr, keys, rank, world_size):
    """Rank-local CSR: rows [u0,u1) with indptr rebased to start at zero.
    Returns (u0, u1, local_indptr, local_keys, nnz_offset)."""
    b = shard_bounds(indptr, world_size)
    u0, u1 = b[rank], b[rank + 1]
    beg = 0 if u0 == 0 else int(indptr[u0 - 1])
    end = beg if u1 == u0 else int(indptr[u1 - 1])
    local_indptr = np.ascontiguousarray(np.asarray(indptr[u0:u1], dtype=np.int64) - beg)
    return u0, u1, local_indptr, np.ascontiguousarray(keys[beg:end]), beg


class DeltaAllReduce:
    """Keeps replicated tensors consistent across ranks: call `begin()` before the local work and
    `finish()` after it.  Tensors are torch tensors aliasing the engine's buffers."""

    def __init__(self, tensors, group=None):
        import torch
        self.torch = torch
        self.tensors = [t for t in tensors if t is not None and t.numel() > 0]
        self.group = group
        self.snap = [torch.empty_like(t) for t in self.tensors]
        self.bytes_per_sync = sum(t.numel() * t.element_size() for t in self.tensors)

    def _sync(self):
        # the backend launches on its own stream and returns idle; torch work on torch's stream has
        # to be complete before the backend touches the same buffers again (and vice versa)
        if self.tensors and self.tensors[0].is_cuda:
            self.torch.cuda.current_stream().synchronize()

    def begin(self):
        for s, t in zip(self.snap, self.tensors):
            s.copy_(t)
        self._sync()

    def finish(self):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return
        for s, t in zip(self.snap, self.tensors):
            t.sub_(s)                                   # local delta
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t.add_(s)                                   # T_sync + sum of deltas
        self._sync()


class PipelinedDeltaExchange:
    """The exchange rule a handle applies by itself once an RCCL rank is attached (`obj.set_comm(Comm(...))`:
    csrc/sgd_base.hip `exchange_begin` / `exchange_finish`), restated on torch tensors so that the protocol runs under
    gloo on CPU.  One exchange is in flight at a time:

        begin():   finish(progressed=True);  S = T - Z;  R = all_reduce(S)  (asynchronous)
        finish():  wait;  Z += R;  T += R - S  if the rank worked on T since begin, else  T = Z

    `Z` (the state every rank agrees on) advances by the same arithmetic everywhere and stays bit-identical; a flush
    (finish without local progress) therefore leaves bit-identical replicas; every local delta is applied exactly once
    on every rank; and between `begin` and `finish` the rank keeps working on its replica -- the all-reduce travels
    behind the next walk instead of in front of it."""

    def __init__(self, tensors, group=None):
        import torch
        self.tensors = [t for t in tensors if t is not None and t.numel() > 0]
        self.group = group
        self.Z = [t.clone() for t in self.tensors]
        self.S = [torch.empty_like(t) for t in self.tensors]
        self.R = [torch.empty_like(t) for t in self.tensors]
        self.work = None

    def begin(self):
        import torch
        import torch.distributed as dist
        self.finish(progressed=True)
        for t, z, s_, r in zip(self.tensors, self.Z, self.S, self.R):
            torch.sub(t, z, out=s_)
            r.copy_(s_)
        self.work = [dist.all_reduce(r, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for r in self.R]

    def finish(self, progressed=False):
        if self.work is None:
            return
        for w in self.work:
            w.wait()
        self.work = None
        for t, z, s_, r in zip(self.tensors, self.Z, self.S, self.R):
            z.add_(r)
            if progressed:
                t.add_(r - s_)
            else:
                t.copy_(z)


class DataParallelSGD:
    """Drives one accelerator object (CyBPR / CyWARP surface) on this rank's user shard, exchanging through
    torch.distributed (the CPU tests with the oracle as the engine; `bench.py` with BFH_COMM=torch).  The product path
    on GPUs is the library's own communicator: `obj.set_comm(Comm(...))` and plain `add_jobs` / `update_parameters`.

    `engine` must offer add_jobs / update_parameters plus `replicated_tensors(kind)` returning the
    torch views to all-reduce; `HipEngine` adapts the HIP backend, the CPU tests plug the oracle in.
    `pipelined` (sgd only) uses PipelinedDeltaExchange -- the rule the library applies -- instead of the blocking
    DeltaAllReduce; call `flush()` before reading the model.
    """

    def __init__(self, engine, optimizer, group=None, pipelined=False):
        self.engine = engine
        self.sgd = optimizer == "sgd"
        self.pipe = PipelinedDeltaExchange(engine.replicated_tensors("model"), group) if (pipelined and self.sgd) else None
        self.sync = None if self.pipe else DeltaAllReduce(engine.replicated_tensors("model" if self.sgd else "grad"), group)

    def minibatch(self, start_x, next_x, indptr, keys):
        """One `add_jobs` over [start_x,next_x) of the local shard + the item-side exchange."""
        if self.pipe is not None:
            out = self.engine.add_jobs(start_x, next_x, indptr, keys)
            self.engine.wait()
            self.pipe.begin()
            return out
        self.sync.begin()
        out = self.engine.add_jobs(start_x, next_x, indptr, keys)
        self.engine.wait()
        self.sync.finish()
        return out

    def flush(self):
        if self.pipe is not None:
            self.pipe.finish()

    def end_epoch(self):
        self.engine.update_parameters()


class HipEngine:
    """Adapter: buffalo_amd.backend.CyBPR / CyWARP -> DataParallelSGD engine."""

    def __init__(self, obj, num_items, vdim, optimizer, pcn=False):
        self.obj, self.I, self.vdim, self.optimizer, self.pcn = obj, num_items, vdim, optimizer, pcn

    def replicated_tensors(self, kind):
        o = self.obj
        if kind == "model":
            return [o.device_tensor("Q", (self.I, self.vdim)), o.device_tensor("Qb", (self.I,))]
        ts = [o.device_tensor("gradQ", (self.I, self.vdim)), o.device_tensor("gradQb", (self.I,))]
        if self.pcn:
            ts.append(o.device_tensor("countQ", (self.I,), dtype="int32"))
        return ts

    def add_jobs(self, start_x, next_x, indptr, keys):
        return self.obj.add_jobs(start_x, next_x, indptr, keys)

    def wait(self):
        import torch
        torch.cuda.synchronize()  # backend calls are synchronous; torch ops run on torch's stream

    def update_parameters(self):
        self.obj.update_parameters()


class DataParallelALS:
    """ALS across ranks: rows inside a half-epoch are independent given the
    other side's factors, so the rows being solved are cut into contiguous nnz-balanced shards, both
    factor matrices are replicated, and after each half-epoch every rank publishes the rows it solved.
    The result is the single-GPU result bit for bit: every row is solved by exactly one rank from
    identical inputs, and FF = F^T F is recomputed by every rank from the (identical) replica -- a
    0.1 ms kernel at ML-20M/d=128, cheaper than all-reducing partial Gramians and free of a second
    summation order.

    Exchange: one broadcast per rank of its contiguous row block (P: 71 MB / world at ML-20M d=128,
    Q: 14 MB / world) -- the uneven-size all-gather written as `world` broadcasts, which RCCL runs as
    direct xGMI copies.  `engine` offers precompute(axis), partial_update(a, b, axis) -> (nume, deno)
    over the FULL-matrix row range [a, b) and factor_tensor(axis) -> torch view [rows, vdim] of the side
    being solved; `HipAlsEngine` adapts CyALS, the CPU tests plug the oracle in."""

    def __init__(self, engine, indptrs, group=None):
        import torch.distributed as dist
        self.engine, self.group = engine, group
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self.bounds = [shard_bounds(ip, self.world) for ip in indptrs]   # [axis] -> world+1 row boundaries

    def half_epoch(self, axis):
        import torch
        self.engine.precompute(axis)
        b = self.bounds[axis]
        loss = self.engine.partial_update(b[self.rank], b[self.rank + 1], axis)
        if self.world == 1:
            return loss
        import torch.distributed as dist
        self.engine.wait()
        F = self.engine.factor_tensor(axis)
        for r in range(self.world):
            if b[r + 1] > b[r]:
                dist.broadcast(F[b[r]:b[r + 1]], src=dist.get_global_rank(self.group, r) if self.group is not None else r, group=self.group)
        l = torch.tensor(loss, dtype=torch.float64, device=F.device)
        dist.all_reduce(l, op=dist.ReduceOp.SUM, group=self.group)
        if F.is_cuda:
            torch.cuda.current_stream().synchronize()
        return float(l[0]), float(l[1])

    def epoch(self):
        """als.py:165-171: rowwise then colwise half-epoch; returns the summed (nume, deno)."""
        n0, d0 = self.half_epoch(0)
        n1, d1 = self.half_epoch(1)
        return n0 + n1, d0 + d1


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


class HipAlsEngine:
    """Adapter: buffalo_amd.backend.CyALS with both CSR orientations resident -> DataParallelALS engine."""

    def __init__(self, obj, num_users, num_items, vdim, lindptr, rindptr):
        self.obj, self.rows, self.vdim = obj, (num_users, num_items), vdim
        self.indptr = (lindptr, rindptr)
        obj.set_mode("als_writeback", 0)      # rows stay in HBM; synchronize(True) copies the model out once

    def precompute(self, axis):
        self.obj.precompute(axis)

    def partial_update(self, a, b, axis):
        return self.obj.partial_update(a, b, self.indptr[axis], None, None, axis)

    def factor_tensor(self, axis):
        return self.obj.device_tensor("P" if axis == 0 else "Q", (self.rows[axis], self.vdim))

    def wait(self):
        import torch
        torch.cuda.synchronize()
