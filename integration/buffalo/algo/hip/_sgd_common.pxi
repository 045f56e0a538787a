# Shared by _bpr.pyx and _warp.pyx (textually included): the accelerator surface of /root/reference/buffalo/algo/cuda/_bpr.pyx:27-80
# over the C ABI of include/buffalo_hip.h.  PREFIX-specific extern declarations live in the including file.

cdef inline _raise(const void* h):
    cdef const char* msg = bfh_last_error(h)
    raise RuntimeError(msg.decode("utf-8", "replace") if msg != NULL else "libbuffalo_hip: unknown error")   # CHECK_CUDA's std::runtime_error -> `except +`
