"""COO -> compressed rows on the GPU (`bfh_coo_to_csr`): the device-side replacement of the sort + binarization
step of buffalo's data creation (/root/reference/buffalo/data/fileio.hpp:263-420, called per orientation from
data/base.py:399-451), and the SPPMI matrix of a stream (`bfh_sppmi_*`: stream.py:257-267 + fileio.hpp:109-254 +
stream.py:169-195).  No CPU fallback."""
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


def _text_bytes(text):
    if isinstance(text, (bytes, bytearray, memoryview)):
        return bytes(text) if not isinstance(text, bytes) else text
    if isinstance(text, np.ndarray) and text.dtype == np.uint8:
        return text.tobytes()
    raise TypeError("text must be bytes (the working file's content) or a uint8 array")


def parse_triples(text, total_lines, with_stats=False):
    """The first `total_lines` lines of buffalo's working text file ("row col val", 1-based ids: data/mm.py:175-234) as the reference's
    sscanf(line, "%d %d %f") reads them (fileio.hpp:300-303), parsed on the device (`bfh_parse_triples`).  Returns (rows, cols, vals) with the
    ids still 1-based; stats["merges"] = lines the device handed back to sscanf."""
    buf = _text_bytes(text)
    n = int(total_lines)
    rows, cols, vals = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.float32)
    st = Stats()
    L = lib()
    rc = L.bfh_parse_triples(buf, len(buf), n, rows.ctypes.data_as(C.POINTER(C.c_int32)), cols.ctypes.data_as(C.POINTER(C.c_int32)),
                             vals.ctypes.data_as(C.POINTER(C.c_float)), C.byref(st))
    if rc < 0:
        raise BuffaloHipError((L.bfh_last_error(None) or b"bfh_parse_triples failed").decode())
    return ((rows, cols, vals), st.as_dict()) if with_stats else (rows, cols, vals)


def text_to_csr(text, total_lines, num_major, num_minor, sort_key, with_stats=False):
    """Working text file -> the `rowwise` (sort_key 1) / `colwise` (sort_key 2) group, everything between the bytes and the group on the device
    (`bfh_text_to_csr` = fileio.hpp:263-420: parse, stable sort by (major, minor), END offsets, 0-based minors)."""
    buf = _text_bytes(text)
    n = int(total_lines)
    indptr = np.empty(int(num_major), dtype=np.int64)
    key, val = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.float32)
    st = Stats()
    L = lib()
    rc = L.bfh_text_to_csr(buf, len(buf), n, int(num_major), int(num_minor), int(sort_key), indptr.ctypes.data_as(C.POINTER(C.c_int64)),
                           key.ctypes.data_as(C.POINTER(C.c_int32)), val.ctypes.data_as(C.POINTER(C.c_float)), C.byref(st))
    if rc < 0:
        raise BuffaloHipError((L.bfh_last_error(None) or b"bfh_text_to_csr failed").decode())
    g = {"indptr": indptr, "key": key, "val": val}
    return (g, st.as_dict()) if with_stats else g


def build_sppmi(indptr, items, num_items, windows, k, with_stats=False):
    """SPPMI group of a stream: `indptr` = END offsets [num_users] over the 0-based `items` of the users' sequences,
    `windows` / `k` = the reference's data.sppmi options (stream.py:34-36).  Returns {"indptr": int64 END offsets
    [num_items], "key": int32 [nnz], "val": float32 [nnz], "total_lines": D} -- the layout of the reference's `sppmi`
    HDF5 group (stream.py:183-188), which CFR reads as its context matrix.  Like the reference's builder, pairs with the
    largest id that occurs are left out (fileio.hpp:182-250 never flushes the group at end of file); rows hold their
    entries in column order (the reference: std::unordered_set iteration order -- same entries per row)."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    items = np.ascontiguousarray(items, dtype=np.int32)
    if indptr.ndim != 1 or items.ndim != 1 or indptr.shape[0] == 0 or int(indptr[-1]) != items.shape[0]:
        raise ValueError("indptr must be 1-d END offsets whose last entry is len(items)")
    L = lib()
    h = L.bfh_sppmi_create()
    if not h:
        raise BuffaloHipError((L.bfh_last_error(None) or b"bfh_sppmi_create failed").decode())
    try:
        nnz, lines = C.c_int64(0), C.c_int64(0)
        rc = L.bfh_sppmi_build(h, indptr.ctypes.data_as(C.POINTER(C.c_int64)), items.ctypes.data_as(C.POINTER(C.c_int32)), indptr.shape[0],
                               int(num_items), int(windows), int(k), C.byref(nnz), C.byref(lines))
        if rc < 0:
            raise BuffaloHipError((L.bfh_last_error(h) or b"bfh_sppmi_build failed").decode())
        out_indptr = np.empty(int(num_items), dtype=np.int64)
        key = np.empty(nnz.value, dtype=np.int32)
        val = np.empty(nnz.value, dtype=np.float32)
        rc = L.bfh_sppmi_fetch(h, out_indptr.ctypes.data_as(C.POINTER(C.c_int64)), key.ctypes.data_as(C.POINTER(C.c_int32)),
                               val.ctypes.data_as(C.POINTER(C.c_float)))
        if rc < 0:
            raise BuffaloHipError((L.bfh_last_error(h) or b"bfh_sppmi_fetch failed").decode())
        st = Stats()
        L.bfh_sppmi_get_stats(h, C.byref(st))
    finally:
        L.bfh_sppmi_destroy(h)
    g = {"indptr": out_indptr, "key": key, "val": val, "total_lines": lines.value}
    return (g, st.as_dict()) if with_stats else g
