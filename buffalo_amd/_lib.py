"""ctypes binding of libbuffalo_hip.so (the C ABI declared in include/buffalo_hip.h).

There is deliberately no fallback: if the shared library is missing or a call fails, an exception
is raised -- the product path never routes through the CPU oracle or any eager substitute.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# BFH_LIBRARY=test selects libbuffalo_hip_test.so: the same objects + the shared-memory TEST transport (csrc/comm_test_transport.hpp) that lets
# N processes share one GPU -- tests and one-GPU rehearsals of bench.py --gpus N only; the product library does not contain it
LIB_PATH = os.path.join(_HERE, "libbuffalo_hip_test.so" if os.environ.get("BFH_LIBRARY") == "test" else "libbuffalo_hip.so")
HEADER_PATH = os.path.join(_HERE, "..", "include", "buffalo_hip.h")


class BuffaloHipError(RuntimeError):
    """Raised for every non-OK status of the C ABI (the reference raises C++ exceptions through
    Cython's `except +`: /root/reference/buffalo/algo/cuda/_bpr.pyx:14-23)."""


class Stats(C.Structure):
    _fields_ = [("samples", C.c_int64), ("scored_negatives", C.c_int64), ("accepted", C.c_int64),
                ("launches", C.c_int64), ("kernel_ms", C.c_double), ("optimizer_ms", C.c_double),
                ("aux_ms", C.c_double), ("h2d_bytes", C.c_double), ("d2h_bytes", C.c_double),
                ("merges", C.c_int64), ("exchanges", C.c_int64), ("loaded_rows", C.c_int64),
                ("exchange_kernel_ms", C.c_double), ("allreduce_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_vp, _i32, _i64, _f64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_size_t
_pf = C.POINTER(C.c_float)
_pi32 = C.POINTER(C.c_int32)
_pi64 = C.POINTER(C.c_int64)
_pf64 = C.POINTER(C.c_double)


def _sgd_sigs(pfx):
    return {
        pfx + "create": (_vp, []),
        pfx + "destroy": (None, [_vp]),
        pfx + "init": (_i32, [_vp, C.c_char_p]),
        pfx + "get_vdim": (_i32, [_vp]),
        pfx + "initialize_model": (_i32, [_vp, _pf, _i32, _pf, _pf, _i32, _i64, _i32]),
        pfx + "set_placeholder": (_i32, [_vp, _pi64, _sz]),
        pfx + "set_cumulative_table": (_i32, [_vp, _pi64]),
        pfx + "partial_update": (_i32, [_vp, _i32, _i32, _pi64, _pi32, _pf64, _pf64]),
        pfx + "update_parameters": (_i32, [_vp]),
        pfx + "synchronize": (_i32, [_vp, _i32]),
        pfx + "compute_loss": (_i32, [_vp, _i32, _pi32, _pi32, _pi32, _pf64]),
        pfx + "set_device": (_i32, [_vp, _i32]),
        pfx + "set_resident_csr": (_i32, [_vp, _pi64, _pi32, _i64]),
        pfx + "set_mode": (_i32, [_vp, C.c_char_p, _i64]),
        pfx + "set_shard": (_i32, [_vp, _i64, _i32]),
        pfx + "device_buffer": (_i32, [_vp, C.c_char_p, C.POINTER(_vp), C.POINTER(_sz)]),
        pfx + "stream": (_vp, [_vp]),
        pfx + "get_stats": (_i32, [_vp, C.POINTER(Stats)]),
        pfx + "reset_stats": (_i32, [_vp]),
        pfx + "set_comm": (_i32, [_vp, _vp]),
        pfx + "comm_flush": (_i32, [_vp]),
    }


SIGNATURES = {
    "bfh_version": (C.c_char_p, []),
    "bfh_stats_size": (C.c_size_t, []),
    "bfh_last_error": (C.c_char_p, [_vp]),
    "bfh_device_count": (_i32, []),
    "bfh_bpr_update_triples": (_i32, [_vp, _i64, _pi32, _pi32, _pi32, _f64]),
    "bfh_bpr_item_major_plan": (_i32, [_i32, _pi64, _i32, _i64, C.POINTER(C.c_int), _pi64, _pi64, _pi64]),
    "bfh_comm_unique_id": (_i32, [C.c_char_p, _sz]),
    "bfh_comm_create": (_vp, [_i32, _i32, C.c_char_p, _i32]),
    "bfh_comm_destroy": (None, [_vp]),
    "bfh_comm_rank": (_i32, [_vp]),
    "bfh_comm_size": (_i32, [_vp]),
    "bfh_comm_transport": (_i32, [_vp, C.c_char_p, C.c_size_t]),
    "bfh_comm_self_test": (_i32, [_vp]),
    "bfh_comm_all_reduce_f64": (_i32, [_vp, _pf64, _i32]),
    "bfh_als_set_comm": (_i32, [_vp, _vp]),
    "bfh_als_publish_rows": (_i32, [_vp, _i32, _pi32, _i32]),
    "bfh_als_create": (_vp, []),
    "bfh_als_destroy": (None, [_vp]),
    "bfh_als_init": (_i32, [_vp, C.c_char_p]),
    "bfh_als_get_vdim": (_i32, [_vp]),
    "bfh_als_initialize_model": (_i32, [_vp, _pf, _i32, _pf, _i32]),
    "bfh_als_set_placeholder": (_i32, [_vp, _pi64, _pi64, _sz]),
    "bfh_als_precompute": (_i32, [_vp, _i32]),
    "bfh_als_partial_update": (_i32, [_vp, _i32, _i32, _pi64, _pi32, _pf, _i32, _pf64, _pf64]),
    "bfh_als_set_device": (_i32, [_vp, _i32]),
    "bfh_als_set_resident_csr": (_i32, [_vp, _i32, _pi64, _pi32, _pf, _i64]),
    "bfh_als_synchronize": (_i32, [_vp, _i32]),
    "bfh_als_set_mode": (_i32, [_vp, C.c_char_p, _i64]),
    "bfh_als_device_buffer": (_i32, [_vp, C.c_char_p, C.POINTER(_vp), C.POINTER(_sz)]),
    "bfh_als_stream": (_vp, [_vp]),
    "bfh_als_get_stats": (_i32, [_vp, C.POINTER(Stats)]),
    "bfh_als_reset_stats": (_i32, [_vp]),
    "bfh_eals_create": (_vp, []),
    "bfh_eals_destroy": (None, [_vp]),
    "bfh_eals_set_device": (_i32, [_vp, _i32]),
    "bfh_eals_init": (_i32, [_vp, C.c_char_p]),
    "bfh_eals_initialize_model": (_i32, [_vp, _pf, _pf, _pf, _i32, _i32]),
    "bfh_eals_precompute_cache": (_i32, [_vp, _i32, C.POINTER(_i64), _pi32, _i32]),
    "bfh_eals_update": (_i32, [_vp, C.POINTER(_i64), _pi32, _pf, _i32]),
    "bfh_eals_estimate_loss": (_i32, [_vp, _i32, C.POINTER(_i64), _pi32, _pf, _i32, _pf, _pf]),
    "bfh_eals_get_stats": (_i32, [_vp, C.POINTER(Stats)]),
    "bfh_eals_reset_stats": (_i32, [_vp]),
    "bfh_cfr_create": (_vp, []),
    "bfh_cfr_destroy": (None, [_vp]),
    "bfh_cfr_set_device": (_i32, [_vp, _i32]),
    "bfh_cfr_init": (_i32, [_vp, C.c_char_p]),
    "bfh_cfr_set_embedding": (_i32, [_vp, _pf, _i32, C.c_char_p]),
    "bfh_cfr_precompute": (_i32, [_vp, C.c_char_p]),
    "bfh_cfr_partial_update_user": (_i32, [_vp, _i32, _i32, C.POINTER(_i64), _pi32, _pf, _pf64]),
    "bfh_cfr_partial_update_item": (_i32, [_vp, _i32, _i32, C.POINTER(_i64), _pi32, _pf, C.POINTER(_i64), _pi32, _pf, _pf64]),
    "bfh_cfr_partial_update_context": (_i32, [_vp, _i32, _i32, C.POINTER(_i64), _pi32, _pf, _pf64]),
    "bfh_cfr_get_stats": (_i32, [_vp, C.POINTER(Stats)]),
    "bfh_cfr_reset_stats": (_i32, [_vp]),
    "bfh_coo_to_csr": (_i32, [_pi32, _pi32, _pf, _i64, _i32, _i32, C.POINTER(_i64), _pi32, _pf, C.POINTER(Stats)]),
    "bfh_parse_triples": (_i32, [C.c_char_p, _i64, _i64, _pi32, _pi32, _pf, C.POINTER(Stats)]),
    "bfh_text_to_csr": (_i32, [C.c_char_p, _i64, _i64, _i32, _i32, _i32, C.POINTER(_i64), _pi32, _pf, C.POINTER(Stats)]),
    "bfh_sppmi_create": (_vp, []),
    "bfh_sppmi_destroy": (None, [_vp]),
    "bfh_sppmi_build": (_i32, [_vp, _pi64, _pi32, _i32, _i32, _i32, _i32, _pi64, _pi64]),
    "bfh_sppmi_fetch": (_i32, [_vp, _pi64, _pi32, _pf]),
    "bfh_sppmi_get_stats": (_i32, [_vp, C.POINTER(Stats)]),
    "bfh_topk_create": (_vp, []),
    "bfh_topk_destroy": (None, [_vp]),
    "bfh_topk_set_device": (_i32, [_vp, _i32]),
    "bfh_topk_dot_topn": (_i32, [_vp, _pi32, _i32, _pf, _i32, _i32, _pf, _i32, _i32, _pf, _i32, _pi32, _pf, _pi32, _i32, _i32]),
    "bfh_topk_dot_topn_device": (_i32, [_vp, _pi32, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _pi32, _pf, _pi32, _i32, _i32]),
    "bfh_topk_quickselect": (_i32, [_vp, _pf, _i32, _i32, _pi32, _i32, _i32]),
    "bfh_topk_set_mode": (_i32, [_vp, C.c_char_p, _i64]),
    "bfh_topk_get_stats": (_i32, [_vp, C.POINTER(Stats)]),
    "bfh_topk_reset_stats": (_i32, [_vp]),
}
SIGNATURES.update(_sgd_sigs("bfh_bpr_"))
SIGNATURES.update(_sgd_sigs("bfh_warp_"))

_lib = None


def header_symbols():
    """Every function name declared in include/buffalo_hip.h."""
    with open(HEADER_PATH) as fin:
        text = fin.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bfh_[a-z0-9_]+)\s*\(", text)))


def _preload_shared_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64.so / libhsa-runtime64.so (SONAME libamdhip64.so.7,
    same as /opt/rocm's).  Two HSA runtimes in one process cannot both open the GPU, so when torch is
    installed its runtime is loaded first and libbuffalo_hip.so's DT_NEEDED libamdhip64.so.7 then
    resolves to it by SONAME; a later `import torch` reuses the same objects.  Without torch the
    system runtime under /opt/rocm/lib is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    d = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(d, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def lib():
    """Load the shared library (no device access happens here)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BuffaloHipError(
            "libbuffalo_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `python -m buffalo_amd._build`. There is no CPU fallback." % LIB_PATH)
    _preload_shared_hip_runtime()
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError here == ABI drift between header and library
        fn.restype = res
        fn.argtypes = args
    if L.bfh_stats_size() != C.sizeof(Stats):   # bfh_*_get_stats writes the library's whole struct into OUR buffer
        raise BuffaloHipError("bfh_stats is %d bytes in libbuffalo_hip.so and %d in buffalo_amd/_lib.py: rebuild / update the mirror"
                              % (L.bfh_stats_size(), C.sizeof(Stats)))
    _lib = L
    return L


def check(handle, status):
    if status is not None and status < 0:
        msg = lib().bfh_last_error(handle)
        raise BuffaloHipError("%s (status %d)" % ((msg or b"").decode("utf-8", "replace"), status))
    return status
