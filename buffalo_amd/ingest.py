"""COO -> compressed rows on the GPU (`bfh_coo_to_csr`): the device-side replacement of the sort + binarization
step of buffalo's data creation (/root/reference/buffalo/data/fileio.hpp:263-420, called per orientation from
data/base.py:399-451).  No CPU fallback."""
import ctypes as C

import numpy as np

from ._lib import BuffaloHipError, Stats, lib


def coo_to_csr(major, minor, vals, num_major, num_minor, with_stats=False):
    """Stable sort by (major, minor), duplicates kept.  Returns {"indptr": int64 END offsets [num_major],
    "key": int32 [nnz], "val": float32 [nnz]} -- the layout of an HDF5 group of the reference
    (`rowwise` / `colwise`: indptr, key, val)."""
    major = np.ascontiguousarray(major, dtype=np.int32)
    minor = np.ascontiguousarray(minor, dtype=np.int32)
    vals = np.ascontiguousarray(vals, dtype=np.float32)
    if not (major.shape == minor.shape == vals.shape and major.ndim == 1):
        raise ValueError("major, minor and vals must be 1-d arrays of one length")
    nnz = major.shape[0]
    indptr = np.empty(int(num_major), dtype=np.int64)
    key = np.empty(nnz, dtype=np.int32)
    val = np.empty(nnz, dtype=np.float32)
    st = Stats()
    L = lib()
    rc = L.bfh_coo_to_csr(major.ctypes.data_as(C.POINTER(C.c_int32)), minor.ctypes.data_as(C.POINTER(C.c_int32)),
                          vals.ctypes.data_as(C.POINTER(C.c_float)), nnz, int(num_major), int(num_minor),
                          indptr.ctypes.data_as(C.POINTER(C.c_int64)), key.ctypes.data_as(C.POINTER(C.c_int32)),
                          val.ctypes.data_as(C.POINTER(C.c_float)), C.byref(st))
    if rc < 0:
        raise BuffaloHipError((L.bfh_last_error(None) or b"bfh_coo_to_csr failed").decode())
    g = {"indptr": indptr, "key": key, "val": val}
    return (g, st.as_dict()) if with_stats else g
