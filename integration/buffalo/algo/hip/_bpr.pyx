# cython: language_level=3, boundscheck=False, wraparound=False
# distutils: language = c
"""buffalo/algo/hip/_bpr.pyx -- the binding a buffalo maintainer adds next to buffalo/algo/cuda/_bpr.pyx: the same `CyBPR` surface
(/root/reference/buffalo/algo/cuda/_bpr.pyx:27-80), bound to libbuffalo_hip.so's C ABI (include/buffalo_hip.h) instead of the CuBPR C++ class."""
cimport numpy as np
from libc.stdint cimport int32_t, int64_t
import numpy as np

np.import_array()

cdef extern from "buffalo_hip.h":
    void* bfh_bpr_create() nogil
    void  bfh_bpr_destroy(void*) nogil
    int   bfh_bpr_init(void*, const char*) nogil
    int   bfh_bpr_get_vdim(void*) nogil
    int   bfh_bpr_initialize_model(void*, float*, int, float*, float*, int, int64_t, int) nogil
    int   bfh_bpr_set_placeholder(void*, const int64_t*, size_t) nogil
    int   bfh_bpr_set_cumulative_table(void*, const int64_t*) nogil
    int   bfh_bpr_partial_update(void*, int, int, const int64_t*, const int32_t*, double*, double*) nogil
    int   bfh_bpr_update_parameters(void*) nogil
    int   bfh_bpr_synchronize(void*, int) nogil
    int   bfh_bpr_compute_loss(void*, int, const int32_t*, const int32_t*, const int32_t*, double*) nogil
    int   bfh_bpr_set_mode(void*, const char*, int64_t) nogil
    const char* bfh_last_error(const void*) nogil

include "_sgd_common.pxi"


cdef class CyBPR:
    """HIP BPRMF object holder (cuda/_bpr.pyx:27-29)"""
    cdef void* obj
    cdef object _keep        # the caller's arrays: the backend stores raw host pointers and writes the model back into them (bpr.cu:334-336)

    def __cinit__(self):
        self.obj = bfh_bpr_create()
        self._keep = {}
        if self.obj == NULL:
            _raise(NULL)

    def __dealloc__(self):
        if self.obj != NULL:
            bfh_bpr_destroy(self.obj)
            self.obj = NULL

    def init(self, opt_path):                                       # cuda/_bpr.pyx:34-35
        cdef bytes b = opt_path if isinstance(opt_path, bytes) else str(opt_path).encode("utf-8")
        cdef int rc = bfh_bpr_init(self.obj, b)
        if rc < 0:
            _raise(self.obj)
        return rc == 1

    def initialize_model(self, np.ndarray[np.float32_t, ndim=2] P, np.ndarray[np.float32_t, ndim=2] Q,
                         np.ndarray[np.float32_t, ndim=2] Qb, int64_t num_nnz, set_gpu=False):   # :37-44
        self._keep.update(P=P, Q=Q, Qb=Qb)
        if bfh_bpr_initialize_model(self.obj, &P[0, 0], <int>P.shape[0], &Q[0, 0], &Qb[0, 0], <int>Q.shape[0], num_nnz, 1 if set_gpu else 0) < 0:
            _raise(self.obj)

    def set_placeholder(self, np.ndarray[np.int64_t, ndim=1] indptr, size_t batch_size):          # :46-47
        if bfh_bpr_set_placeholder(self.obj, <const int64_t*>&indptr[0], batch_size) < 0:
            _raise(self.obj)

    def set_cumulative_table(self, np.ndarray[np.int64_t, ndim=1] sampling_table, size):          # :49-50
        self._keep["cum"] = sampling_table
        if bfh_bpr_set_cumulative_table(self.obj, <const int64_t*>&sampling_table[0]) < 0:
            _raise(self.obj)

    def get_vdim(self):                                             # :52-53
        return bfh_bpr_get_vdim(self.obj)

    def synchronize(self, device_to_host):                          # :55-57
        if bfh_bpr_synchronize(self.obj, 1 if device_to_host else 0) < 0:
            _raise(self.obj)

    def update_parameters(self):                                    # :59-61 is synchronize(True) only; here adam / adagrad also step on the device
        if bfh_bpr_update_parameters(self.obj) < 0:
            _raise(self.obj)
        self.synchronize(True)

    def wait_until_done(self):                                      # :63-64
        return

    def add_jobs(self, int start_x, int next_x, np.ndarray[np.int64_t, ndim=1] indptr, np.ndarray[np.int32_t, ndim=1] keys):   # :66-74
        cdef double loss = 0, n = 0
        if bfh_bpr_partial_update(self.obj, start_x, next_x, <const int64_t*>&indptr[0], <const int32_t*>&keys[0] if keys is not None and keys.shape[0] else <const int32_t*>NULL, &loss, &n) < 0:
            _raise(self.obj)
        return loss, n

    def compute_loss(self, np.ndarray[np.int32_t, ndim=1] user, np.ndarray[np.int32_t, ndim=1] pos,
                     np.ndarray[np.int32_t, ndim=1] neg):           # :76-80
        cdef double out = 0
        if bfh_bpr_compute_loss(self.obj, <int>user.shape[0], <const int32_t*>&user[0], <const int32_t*>&pos[0], <const int32_t*>&neg[0], &out) < 0:
            _raise(self.obj)
        return out

    def set_mode(self, name, int64_t value):                        # extension (documented in include/buffalo_hip.h): backend knobs
        cdef bytes b = name if isinstance(name, bytes) else str(name).encode("utf-8")
        if bfh_bpr_set_mode(self.obj, b, value) < 0:
            _raise(self.obj)
