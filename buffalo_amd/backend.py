"""Accelerator objects with the method surface of buffalo's Cython CUDA bindings, driving
libbuffalo_hip.so through its C ABI.

* ``CyBPR``  mirrors ``buffalo.algo.cuda._bpr.CyBPR``  (/root/reference/buffalo/algo/cuda/_bpr.pyx:27-80)
* ``CyALS``  mirrors ``buffalo.algo.cuda._als.CyALS``  (/root/reference/buffalo/algo/cuda/_als.pyx:25-67)
* ``CyCFR``  mirrors ``buffalo.algo._cfr.CyCFR``       (/root/reference/buffalo/algo/_cfr.pyx:25-71; CPU layout: [rows, d] unpadded)
* ``CyWARP`` gives WARP the same surface as CyBPR, which is what the (unreachable) accelerator
  scaffold in /root/reference/buffalo/algo/warp.py:212-234 expects.

Same names, argument order and meaning; numpy arrays are handed over as raw pointers and must stay
alive while the object uses them (the bindings keep references, like the reference's callers do).
Failures raise ``BuffaloHipError`` where the reference raises the C++ exception through Cython.
"""
import ctypes as C

import numpy as np

from ._lib import BuffaloHipError, Stats, check, lib


def _arr(a, dtype, ndim, name):
    # Cython's typed buffer arguments raise ValueError on dtype / ndim mismatch
    if not isinstance(a, np.ndarray):
        raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray, got %s)" % (name, type(a).__name__))
    if a.dtype != dtype:
        raise ValueError("Buffer dtype mismatch for '%s', expected '%s' but got '%s'" % (name, np.dtype(dtype), a.dtype))
    if a.ndim != ndim:
        raise ValueError("Buffer has wrong number of dimensions for '%s' (expected %d, got %d)" % (name, ndim, a.ndim))
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("ndarray '%s' is not C-contiguous" % name)
    return a


def _ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _path(p):
    return p if isinstance(p, bytes) else str(p).encode("utf-8")


class _DeviceView:
    """__cuda_array_interface__ carrier so torch can alias a backend buffer without a copy."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self._owner = owner


class Comm:
    """One RCCL rank (`bfh_comm_*`): the multi-GPU plumbing inside the library.  Rank 0 makes the id with
    `Comm.unique_id()` and hands the 128 bytes to the other ranks (any side channel); every rank then builds
    `Comm(world, rank, uid, device)` and attaches it with `obj.set_comm(comm)`.  Keep it alive as long as the objects."""

    def __init__(self, world, rank, uid, device):
        self._L = lib()
        if len(uid) != 128:
            raise ValueError("the RCCL unique id is 128 bytes")
        self._h = self._L.bfh_comm_create(int(world), int(rank), bytes(uid), int(device))
        if not self._h:
            raise BuffaloHipError((self._L.bfh_last_error(None) or b"comm_create failed").decode())
        self.world, self.rank = int(world), int(rank)

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        L = lib()
        if L.bfh_comm_unique_id(buf, 128) < 0:
            raise BuffaloHipError((L.bfh_last_error(None) or b"comm_unique_id failed").decode())
        return buf.raw

    def self_test(self):
        check(self._h, self._L.bfh_comm_self_test(self._h))

    def size(self):
        """Ranks the LIVE communicator reports (ncclCommCount) -- not the number it was asked for."""
        n = self._L.bfh_comm_size(self._h)
        if n < 0:
            check(self._h, n)
        return int(n)

    def transport(self):
        """"rccl <version>", or "shm-test" for libbuffalo_hip_test.so's shared-memory transport."""
        buf = C.create_string_buffer(64)
        check(self._h, self._L.bfh_comm_transport(self._h, buf, 64))
        return buf.value.decode()

    def all_reduce(self, values):
        """Sum of a small list of host doubles over the ranks (loss sums)."""
        arr = (C.c_double * len(values))(*[float(v) for v in values])
        check(self._h, self._L.bfh_comm_all_reduce_f64(self._h, arr, len(values)))
        return [float(v) for v in arr]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.bfh_comm_destroy(self._h)
                self._h = None
        except Exception:
            pass


class _Base:
    _PFX = None

    def __init__(self):
        self._L = lib()
        self._h = getattr(self._L, self._PFX + "create")()
        if not self._h:
            raise BuffaloHipError((self._L.bfh_last_error(None) or b"create failed").decode())
        self._keep = {}

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                getattr(self._L, self._PFX + "destroy")(self._h)
                self._h = None
        except Exception:
            pass

    def _call(self, name, *args):
        return check(self._h, getattr(self._L, self._PFX + name)(self._h, *args))

    def init(self, opt_path):
        return bool(self._call("init", _path(opt_path)))

    def get_vdim(self):
        return self._call("get_vdim")

    # ---- extensions ---------------------------------------------------------------------------
    def set_device(self, device):
        self._call("set_device", int(device))

    def set_mode(self, name, value):
        self._call("set_mode", name.encode(), int(value))

    def stats(self):
        s = Stats()
        self._call("get_stats", C.byref(s))
        return s.as_dict()

    def reset_stats(self):
        self._call("reset_stats")

    def stream(self):
        return getattr(self._L, self._PFX + "stream")(self._h)

    def set_comm(self, comm):
        """Attach an RCCL rank (`Comm`); None detaches.  From then on the object exchanges with its peers by itself."""
        self._call("set_comm", comm._h if comm is not None else None)   # finishes an exchange in flight on the OLD communicator
        self._keep["comm"] = comm                                         # ... which may only go away after that

    def device_buffer(self, name):
        p, n = C.c_void_p(), C.c_size_t()
        self._call("device_buffer", name.encode(), C.byref(p), C.byref(n))
        return p.value, n.value

    def device_tensor(self, name, shape=None, dtype="float32"):
        """Zero-copy torch view of a backend buffer (for RCCL collectives via torch.distributed)."""
        import torch
        ptr, nbytes = self.device_buffer(name)
        if not ptr:
            raise BuffaloHipError("device buffer '%s' is not allocated" % name)
        item = np.dtype(dtype).itemsize
        if shape is None:
            shape = (nbytes // item,)
        typestr = {"float32": "<f4", "int32": "<i4"}[str(np.dtype(dtype))]
        return torch.as_tensor(_DeviceView(ptr, shape, typestr, self), device="cuda")


class _SgdBase(_Base):
    """Shared by CyBPR and CyWARP (accelerator surface of cuda/_bpr.pyx)."""
    # drop-in semantics: the reference copies P,Q,Qb back to the numpy arrays after every epoch
    # (`update_parameters -> synchronize(True)`, cuda/_bpr.pyx:59-61).  Set False to keep the model
    # in HBM until `synchronize(True)` is called explicitly.
    sync_every_epoch = True

    def initialize_model(self, P, Q, Qb, num_nnz, set_gpu=False):
        _arr(P, np.float32, 2, "P"), _arr(Q, np.float32, 2, "Q"), _arr(Qb, np.float32, 2, "Qb")
        if set_gpu:
            vdim = self.get_vdim()
            if P.shape[1] != vdim or Q.shape[1] != vdim:
                raise ValueError("factor matrices must be padded to vdim=%d columns (got %d / %d)"
                                 % (vdim, P.shape[1], Q.shape[1]))
        if Qb.shape[0] != Q.shape[0]:
            raise ValueError("Qb must have one row per item")
        self._keep.update(P=P, Q=Q, Qb=Qb)
        self._call("initialize_model", _ptr(P, C.c_float), P.shape[0], _ptr(Q, C.c_float), _ptr(Qb, C.c_float),
                   Q.shape[0], int(num_nnz), int(bool(set_gpu)))

    def set_placeholder(self, indptr, batch_size):
        _arr(indptr, np.int64, 1, "indptr")
        self._call("set_placeholder", _ptr(indptr, C.c_int64), int(batch_size))

    def set_cumulative_table(self, sampling_table, size):
        _arr(sampling_table, np.int64, 1, "sampling_table")
        self._keep["cum"] = (sampling_table, int(size))
        if "Q" in self._keep:
            # the CUDA binding passes only the pointer; Q_rows entries are read (bpr.cu:322-327)
            if sampling_table.shape[0] < self._keep["Q"].shape[0]:
                raise ValueError("sampling_table shorter than the number of items")
            self._call("set_cumulative_table", _ptr(sampling_table, C.c_int64))

    def synchronize(self, device_to_host):
        self._call("synchronize", int(bool(device_to_host)))

    def flush_host(self):
        """With `set_mode("lazy_sync", 1)` a `synchronize(True)` only marks the numpy arrays stale; this copies now."""
        self._call("synchronize", 2)

    def update_parameters(self):
        self._call("update_parameters")
        if self.sync_every_epoch:
            self.synchronize(True)

    def wait_until_done(self):
        return

    def add_jobs(self, start_x, next_x, indptr, keys):
        _arr(indptr, np.int64, 1, "indptr")
        loss, n = C.c_double(0), C.c_double(0)
        kp = None
        if keys is not None:
            _arr(keys, np.int32, 1, "keys")
            kp = _ptr(keys, C.c_int32)
        self._call("partial_update", int(start_x), int(next_x), _ptr(indptr, C.c_int64), kp, C.byref(loss), C.byref(n))
        return loss.value, n.value

    def compute_loss(self, user, pos, neg):
        _arr(user, np.int32, 1, "user"), _arr(pos, np.int32, 1, "pos"), _arr(neg, np.int32, 1, "neg")
        out = C.c_double(0)
        self._call("compute_loss", user.shape[0], _ptr(user, C.c_int32), _ptr(pos, C.c_int32), _ptr(neg, C.c_int32),
                   C.byref(out))
        return out.value

    # ---- extensions ---------------------------------------------------------------------------
    def set_resident_csr(self, indptr, keys):
        _arr(indptr, np.int64, 1, "indptr"), _arr(keys, np.int32, 1, "keys")
        self._call("set_resident_csr", _ptr(indptr, C.c_int64), _ptr(keys, C.c_int32), int(keys.shape[0]))

    def set_shard(self, nnz_offset, num_shards):
        self._call("set_shard", int(nnz_offset), int(num_shards))

    def comm_flush(self):
        """Finish the exchange that is still in flight (the model then holds every rank's updates)."""
        self._call("comm_flush")


class CyBPR(_SgdBase):
    _PFX = "bfh_bpr_"

    def update_triples(self, users, pos, neg, lr):
        _arr(users, np.int32, 1, "users"), _arr(pos, np.int32, 1, "pos"), _arr(neg, np.int32, 1, "neg")
        check(self._h, self._L.bfh_bpr_update_triples(self._h, users.shape[0], _ptr(users, C.c_int32),
                                                      _ptr(pos, C.c_int32), _ptr(neg, C.c_int32), float(lr)))


class CyWARP(_SgdBase):
    _PFX = "bfh_warp_"


class CyALS(_Base):
    _PFX = "bfh_als_"

    def initialize_model(self, P, Q):
        _arr(P, np.float32, 2, "P"), _arr(Q, np.float32, 2, "Q")
        vdim = self.get_vdim()
        if P.shape[1] != vdim or Q.shape[1] != vdim:
            raise ValueError("factor matrices must have vdim=%d columns (got %d / %d)" % (vdim, P.shape[1], Q.shape[1]))
        self._keep.update(P=P, Q=Q)
        self._call("initialize_model", _ptr(P, C.c_float), P.shape[0], _ptr(Q, C.c_float), Q.shape[0])

    def set_placeholder(self, lindptr, rindptr, batch_size):
        _arr(lindptr, np.int64, 1, "lindptr"), _arr(rindptr, np.int64, 1, "rindptr")
        self._call("set_placeholder", _ptr(lindptr, C.c_int64), _ptr(rindptr, C.c_int64), int(batch_size))

    def precompute(self, axis):
        self._call("precompute", int(axis))

    def partial_update(self, start_x, next_x, indptr, keys, vals, axis):
        _arr(indptr, np.int64, 1, "indptr")
        kp = vp = None
        if keys is not None:
            _arr(keys, np.int32, 1, "keys"), _arr(vals, np.float32, 1, "vals")
            kp, vp = _ptr(keys, C.c_int32), _ptr(vals, C.c_float)
        nume, deno = C.c_double(0), C.c_double(0)
        self._call("partial_update", int(start_x), int(next_x), _ptr(indptr, C.c_int64), kp, vp, int(axis),
                   C.byref(nume), C.byref(deno))
        return nume.value, deno.value

    # ---- extensions ---------------------------------------------------------------------------
    def set_resident_csr(self, axis, indptr, keys, vals):
        _arr(indptr, np.int64, 1, "indptr"), _arr(keys, np.int32, 1, "keys"), _arr(vals, np.float32, 1, "vals")
        self._call("set_resident_csr", int(axis), _ptr(indptr, C.c_int64), _ptr(keys, C.c_int32),
                   _ptr(vals, C.c_float), int(keys.shape[0]))

    def synchronize(self, device_to_host):
        self._call("synchronize", int(bool(device_to_host)))

    def publish_rows(self, axis, bounds):
        """Multi-GPU: after `partial_update` on this rank's rows of `axis`, exchange the solved row blocks
        (`bounds` = world_size + 1 row boundaries, the same on every rank)."""
        b = np.ascontiguousarray(bounds, dtype=np.int32)
        self._call("publish_rows", int(axis), _ptr(b, C.c_int32), int(b.shape[0]))


class CyCFR(_Base):
    """CoFactor row updates on the ALS Gramian / dense-solve kernels (csrc/cfr_impl.hpp)."""
    _PFX = "bfh_cfr_"

    def set_embedding(self, F, obj_type):
        _arr(F, np.float32, 2, "F")
        t = obj_type if isinstance(obj_type, bytes) else str(obj_type).encode("utf-8")
        self._keep[t] = F
        self._call("set_embedding", _ptr(F, C.c_float), F.shape[0], t)

    def precompute(self, obj_type):
        self._call("precompute", obj_type if isinstance(obj_type, bytes) else str(obj_type).encode("utf-8"))

    def partial_update_user(self, start_x, next_x, indptrs, keys, vals):
        _arr(indptrs, np.int64, 1, "indptrs"), _arr(keys, np.int32, 1, "keys"), _arr(vals, np.float32, 1, "vals")
        loss = C.c_double(0.0)
        self._call("partial_update_user", int(start_x), int(next_x), _ptr(indptrs, C.c_int64), _ptr(keys, C.c_int32), _ptr(vals, C.c_float),
                   C.byref(loss))
        return loss.value

    def partial_update_item(self, start_x, next_x, indptrs_u, keys_u, vals_u, indptrs_c, keys_c, vals_c):
        for a, dt, n in ((indptrs_u, np.int64, "indptrs_u"), (keys_u, np.int32, "keys_u"), (vals_u, np.float32, "vals_u"),
                         (indptrs_c, np.int64, "indptrs_c"), (keys_c, np.int32, "keys_c"), (vals_c, np.float32, "vals_c")):
            _arr(a, dt, 1, n)
        loss = C.c_double(0.0)
        self._call("partial_update_item", int(start_x), int(next_x), _ptr(indptrs_u, C.c_int64), _ptr(keys_u, C.c_int32), _ptr(vals_u, C.c_float),
                   _ptr(indptrs_c, C.c_int64), _ptr(keys_c, C.c_int32), _ptr(vals_c, C.c_float), C.byref(loss))
        return loss.value

    def partial_update_context(self, start_x, next_x, indptrs, keys, vals):
        _arr(indptrs, np.int64, 1, "indptrs"), _arr(keys, np.int32, 1, "keys"), _arr(vals, np.float32, 1, "vals")
        loss = C.c_double(0.0)
        self._call("partial_update_context", int(start_x), int(next_x), _ptr(indptrs, C.c_int64), _ptr(keys, C.c_int32), _ptr(vals, C.c_float),
                   C.byref(loss))
        return loss.value


class CyEALS(_Base):
    """Element-wise ALS (csrc/eals_impl.hpp); mirrors buffalo.algo._eals.CyEALS (/root/reference/buffalo/algo/_eals.pyx:23-67)."""
    _PFX = "bfh_eals_"

    def initialize_model(self, P, Q, Cw):
        _arr(P, np.float32, 2, "P"), _arr(Q, np.float32, 2, "Q"), _arr(Cw, np.float32, 1, "C")
        self._keep.update(P=P, Q=Q, C=Cw)
        self._call("initialize_model", _ptr(P, C.c_float), _ptr(Q, C.c_float), _ptr(Cw, C.c_float), P.shape[0], Q.shape[0])

    def precompute_cache(self, nnz, indptr, keys, axis):
        _arr(indptr, np.int64, 1, "indptr"), _arr(keys, np.int32, 1, "keys")
        self._call("precompute_cache", int(nnz), _ptr(indptr, C.c_int64), _ptr(keys, C.c_int32), int(axis))

    def update(self, indptr, keys, vals, axis):
        _arr(indptr, np.int64, 1, "indptr"), _arr(keys, np.int32, 1, "keys"), _arr(vals, np.float32, 1, "vals")
        return bool(self._call("update", _ptr(indptr, C.c_int64), _ptr(keys, C.c_int32), _ptr(vals, C.c_float), int(axis)))

    def estimate_loss(self, nnz, indptr, keys, vals, axis):
        _arr(indptr, np.int64, 1, "indptr"), _arr(keys, np.int32, 1, "keys"), _arr(vals, np.float32, 1, "vals")
        rmse, loss = C.c_float(0.0), C.c_float(0.0)
        self._call("estimate_loss", int(nnz), _ptr(indptr, C.c_int64), _ptr(keys, C.c_int32), _ptr(vals, C.c_float), int(axis),
                   C.byref(rmse), C.byref(loss))
        return rmse.value, loss.value
