"""The exchange protocol of DESIGN.md section 8 RESTATED on torch tensors (TEST INFRASTRUCTURE -- not on the product path).

The product implementation is the library's own (csrc/sgd_base.hip exchange_arm / _begin / _finish / exchange_gradients,
csrc/als_kernels.hpp publish_rows over csrc/comm.hip); it runs with N > 1 ranks on one GPU in tests/test_comm_ranks_gpu.py.
These classes exist so that the SAME protocol -- delta all-reduce, the one-deep pipelined exchange, row publishing -- runs under
gloo on CPU with the oracle as the engine (tests/test_dist_cpu.py: world 2, including a model-quality test), where no GPU exists.
"""
import numpy as np  # noqa: F401

from buffalo_amd.dist import shard_bounds, shard_csr  # noqa: F401


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
    torch.distributed (the CPU tests with the oracle as the engine; `bench.py` has had no such path since round 6: without the library's communicator it fails).  The product path
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
