# cython: language_level=3, boundscheck=False, wraparound=False
# distutils: language = c
"""buffalo/algo/hip/_als.pyx -- the `CyALS` surface of /root/reference/buffalo/algo/cuda/_als.pyx:25-67 bound to libbuffalo_hip.so's
C ABI (include/buffalo_hip.h) instead of the CuALS C++ class."""
cimport numpy as np
from libc.stdint cimport int32_t, int64_t
import numpy as np

np.import_array()

cdef extern from "buffalo_hip.h":
    void* bfh_als_create() nogil
    void  bfh_als_destroy(void*) nogil
    int   bfh_als_init(void*, const char*) nogil
    int   bfh_als_get_vdim(void*) nogil
    int   bfh_als_initialize_model(void*, float*, int, float*, int) nogil
    int   bfh_als_set_placeholder(void*, const int64_t*, const int64_t*, size_t) nogil
    int   bfh_als_precompute(void*, int) nogil
    int   bfh_als_partial_update(void*, int, int, const int64_t*, const int32_t*, const float*, int, double*, double*) nogil
    int   bfh_als_set_mode(void*, const char*, int64_t) nogil
    const char* bfh_last_error(const void*) nogil

cdef inline _raise(const void* h):
    cdef const char* msg = bfh_last_error(h)
    raise RuntimeError(msg.decode("utf-8", "replace") if msg != NULL else "libbuffalo_hip: unknown error")


cdef class CyALS:
    """HIP ALS object holder (cuda/_als.pyx:25-27)"""
    cdef void* obj
    cdef object _keep        # the updated rows are written back into the caller's arrays after every partial_update (als.cu:403)

    def __cinit__(self):
        self.obj = bfh_als_create()
        self._keep = {}
        if self.obj == NULL:
            _raise(NULL)

    def __dealloc__(self):
        if self.obj != NULL:
            bfh_als_destroy(self.obj)
            self.obj = NULL

    def init(self, opt_path):                                       # :35-36
        cdef bytes b = opt_path if isinstance(opt_path, bytes) else str(opt_path).encode("utf-8")
        cdef int rc = bfh_als_init(self.obj, b)
        if rc < 0:
            _raise(self.obj)
        return rc == 1

    def initialize_model(self, np.ndarray[np.float32_t, ndim=2] P, np.ndarray[np.float32_t, ndim=2] Q):   # :38-42
        self._keep.update(P=P, Q=Q)
        if bfh_als_initialize_model(self.obj, &P[0, 0], <int>P.shape[0], &Q[0, 0], <int>Q.shape[0]) < 0:
            _raise(self.obj)

    def set_placeholder(self, np.ndarray[np.int64_t, ndim=1] lindptr, np.ndarray[np.int64_t, ndim=1] rindptr, size_t batch_size):   # :44-47
        if bfh_als_set_placeholder(self.obj, <const int64_t*>&lindptr[0], <const int64_t*>&rindptr[0], batch_size) < 0:
            _raise(self.obj)

    def precompute(self, axis):                                     # :49-50
        if bfh_als_precompute(self.obj, axis) < 0:
            _raise(self.obj)

    def get_vdim(self):                                             # :52-53
        return bfh_als_get_vdim(self.obj)

    def partial_update(self, int start_x, int next_x, np.ndarray[np.int64_t, ndim=1] indptr, np.ndarray[np.int32_t, ndim=1] keys,
                       np.ndarray[np.float32_t, ndim=1] vals, int axis):   # :55-67 (returns the (loss_nume, loss_deno) pair)
        cdef double nume = 0, deno = 0
        cdef bint have = keys is not None and vals is not None and keys.shape[0] > 0
        if bfh_als_partial_update(self.obj, start_x, next_x, <const int64_t*>&indptr[0], <const int32_t*>&keys[0] if have else <const int32_t*>NULL, <const float*>&vals[0] if have else <const float*>NULL, axis, &nume, &deno) < 0:
            _raise(self.obj)
        return nume, deno

    def set_mode(self, name, int64_t value):                        # extension: backend knobs
        cdef bytes b = name if isinstance(name, bytes) else str(name).encode("utf-8")
        if bfh_als_set_mode(self.obj, b, value) < 0:
            _raise(self.obj)
