"""The REFERENCE's own data-ingestion C++ behind ctypes (TEST INFRASTRUCTURE ONLY).

oracle/_ref/libbuffalo_fileio_ref.so is /root/reference/buffalo/data/fileio.hpp compiled from where it lies behind the C door of
oracle/ref_fileio.cc (oracle/Makefile target `_ref`; nothing of the reference is copied into this repository, and oracle/_ref/
is git-ignored).  The functions below drive it exactly as the reference's Python does -- text files in, binary files out -- and
hand back numpy arrays, so that tests can pin the oracle's restatements (oracle.coo_to_csr, oracle.build_sppmi) against the
real thing.  `available()` is False on machines that have neither /root/reference nor a prebuilt library (the tests then fall
back to the golden vectors this module generated: tests/golden/make_fileio_vectors.py).
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_ref", "libbuffalo_fileio_ref.so")
_REFERENCE = os.environ.get("BUFFALO_REFERENCE", "/root/reference")
_lib = None


def reference_present():
    return os.path.exists(os.path.join(_REFERENCE, "buffalo", "data", "fileio.hpp"))


def build(force=False):
    """make -C oracle _ref (needs the reference tree); returns the library path or None."""
    if not reference_present():
        return _LIB_PATH if os.path.exists(_LIB_PATH) else None
    src = os.path.join(_HERE, "ref_fileio.cc")
    if not force and os.path.exists(_LIB_PATH) and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "_ref", f"REFERENCE={_REFERENCE}"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def available():
    return os.path.exists(_LIB_PATH) or reference_present()


def lib():
    global _lib
    if _lib is None:
        if build() is None:
            raise RuntimeError("oracle/_ref/libbuffalo_fileio_ref.so is not built and /root/reference is absent")
        L = C.CDLL(_LIB_PATH)
        L.ref_parallel_build_sppmi.restype = C.c_longlong
        L.ref_parallel_build_sppmi.argtypes = [C.c_char_p, C.c_char_p, C.c_longlong, C.c_int, C.c_int, C.c_int]
        L.ref_sort_and_compressed_binarization.restype = C.c_int
        L.ref_sort_and_compressed_binarization.argtypes = [C.c_char_p, C.c_char_p, C.c_longlong, C.c_int, C.c_int, C.c_int]
        L.ref_chunking_into_bins.restype = C.c_int
        L.ref_chunking_into_bins.argtypes = [C.c_char_p, C.c_char_p, C.c_longlong, C.c_int, C.c_int, C.c_int]
        _lib = L
    return _lib


def _fmt_val(v):
    """Values are written the way the reference's readers write them: repr of the Python number (mm.py / stream.py f-strings)."""
    return repr(float(v)) if float(v) != int(v) else str(int(v))


def write_triples(path, rows, cols, vals):
    """The working file of Data._create_working_data: one "row col val" line per entry, 1-based ids."""
    with open(path, "w") as f:
        for r, c, v in zip(rows.tolist(), cols.tolist(), vals.tolist()):
            f.write(f"{r + 1} {c + 1} {_fmt_val(v)}\n")


def sort_and_compressed_binarization(rows, cols, vals, max_key, sort_key, num_workers=2):
    """Data._sort_and_compressed_binarization + _load_compressed_triplet_bin (data/base.py:292-339 over fileio.hpp:263-420):
    0-based COO in, (indptr END offsets, key, val) of the side `sort_key` names out (1 rowwise, 2 colwise, -1 keep order)."""
    rows = np.asarray(rows); cols = np.asarray(cols); vals = np.asarray(vals)
    n = int(rows.shape[0])
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "working.txt")
        write_triples(src, rows, cols, vals)
        nfiles = lib().ref_sort_and_compressed_binarization(src.encode(), d.encode(), n, int(max_key), int(sort_key), int(num_workers))
        assert nfiles == num_workers + 1
        indptr = np.fromfile(os.path.join(d, "indptr.bin"), dtype=np.int64)
        assert indptr.shape[0] == max_key, (indptr.shape, max_key)   # base.py:311
        rec = np.dtype([("i", "<i4"), ("v", "<f4")])
        parts = [np.fromfile(os.path.join(d, f"chunk{i}.bin"), dtype=rec) for i in range(num_workers)]
    data = np.concatenate(parts)
    assert data.shape[0] == n   # base.py:336
    return {"indptr": indptr, "key": data["i"].copy(), "val": data["v"].copy()}


def _psort_first_field(path, key=1):
    """aux.psort (buffalo/misc/_aux.py:111-131): `sort -n -s -t" " -k <key>` under LC_ALL=C -- numeric and STABLE on the key-th
    field.  The very command when sort(1) is on PATH, else Python's stable sort (the two are compared in the tests)."""
    import shutil
    if shutil.which("sort"):
        subprocess.check_call(["sort", "-n", "-s", "-t", " ", "-k", str(key), "-o", path, path], env={"LC_ALL": "C", "PATH": os.environ.get("PATH", "")})
        return
    _psort_python(path, key)


def _psort_python(path, key=1):
    with open(path) as f:
        lines = f.readlines()
    lines.sort(key=lambda s: int(s.split(" ")[key - 1]))
    with open(path, "w") as f:
        f.writelines(lines)


def pair_lines(indptr, items, windows):
    """StreamData._create_working_data's SPPMI lines (stream.py:257-267): for every user's sequence, each item and the `windows`
    items after it as the two lines "w c" and "c w" (1-based)."""
    out = []
    beg = 0
    for end in np.asarray(indptr).tolist():
        seq = (np.asarray(items[beg:end]) + 1).tolist()
        sz = len(seq)
        for i in range(sz):
            for j in range(i + 1, min(i + windows + 1, sz)):
                out.append(f"{seq[i]} {seq[j]}\n")
                out.append(f"{seq[j]} {seq[i]}\n")
        beg = end
    return out


def build_sppmi(indptr, items, num_items, windows, k, num_workers=1, num_chunks=4, return_lines=False):
    """StreamData._build_sppmi (stream.py:169-195) around the compiled reference: pair lines -> psort(key=1) ->
    _parallel_build_sppmi (fileio.hpp:109-254) -> psort -> _chunking_into_bins (fileio.hpp:25-107) ->
    _build_compressed_triplets (data/base.py:354-398, restated here with numpy: it is h5py bookkeeping around np.frombuffer).
    Returns the sppmi group (indptr END offsets over num_items rows, key, val); within a row the entries are in the order the
    reference leaves them (psort is stable on the row id only), which depends on std::unordered_set iteration order."""
    lines = pair_lines(indptr, items, windows)
    total_lines = len(lines)
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "pairs.txt")
        dst = os.path.join(d, "sppmi.txt")
        with open(src, "w") as f:
            f.writelines(lines)
        _psort_first_field(src, 1)
        nnz = int(lib().ref_parallel_build_sppmi(src.encode(), dst.encode(), total_lines, int(num_items), int(k), int(num_workers)))
        _psort_first_field(dst, 1)
        with open(dst) as f:
            text = f.readlines()
        assert len(text) == nnz, (len(text), nnz)
        out_indptr = np.zeros(int(num_items), dtype=np.int64)
        key = np.empty(nnz, dtype=np.int32)
        val = np.empty(nnz, dtype=np.float32)
        if nnz:
            nfiles = lib().ref_chunking_into_bins(dst.encode(), d.encode(), nnz, int(num_chunks), 0, int(num_workers))
            assert nfiles == num_chunks
            rec = np.dtype([("u", "<i4"), ("i", "<i4"), ("v", "<f4")])
            indptr_index = data_index = prev_key = 0
            for ci in range(num_chunks):   # base.py:361-395
                data = np.fromfile(os.path.join(d, f"chunk{ci}.bin"), dtype=rec)
                if data.shape[0] == 0:
                    continue
                U, I, V = data["u"], data["i"], data["v"]
                n = data.shape[0]
                key[data_index:data_index + n] = I
                val[data_index:data_index + n] = V
                diff = U[1:] - U[:-1]
                max_diff = int(np.amax(diff)) if len(diff) else 0
                ip = [data_index for _ in range(int(U[0]) - prev_key)]
                for i in range(max_diff):
                    ip += (np.where(diff > i)[0] + data_index + 1).tolist()
                ip.sort()
                out_indptr[indptr_index:indptr_index + len(ip)] = ip
                data_index += n
                indptr_index += len(ip)
                prev_key = int(U[-1])
            out_indptr[indptr_index:] = data_index
    res = {"indptr": out_indptr, "key": key, "val": val, "total_lines": total_lines}
    if return_lines:
        res["text"] = text
    return res


def timed_sort_and_compressed_binarization(rows, cols, vals, max_key, sort_key, num_workers):
    """Wall time of the reference's compiled `_sort_and_compressed_binarization` (fileio.hpp:263-420: parse the working text file in
    4 MiB splits, stable parallel sort, indptr, binary chunks) on 0-based COO records; the file is written here, untimed."""
    import time
    n = int(np.asarray(rows).shape[0])
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "working.txt")
        np.savetxt(src, np.stack([np.asarray(rows, np.float64) + 1, np.asarray(cols, np.float64) + 1, np.asarray(vals, np.float64)], 1), fmt="%d %d %g")
        t0 = time.perf_counter()
        files = lib().ref_sort_and_compressed_binarization(src.encode(), d.encode(), n, int(max_key), int(sort_key), int(num_workers))
        dt = time.perf_counter() - t0
        assert files == num_workers + 1
        indptr = np.fromfile(os.path.join(d, "indptr.bin"), dtype=np.int64)
    return {"records": n, "workers": int(num_workers), "total_s": dt, "indptr": indptr}


def timed_build_sppmi(indptr, items, num_items, windows, k, num_workers):
    """Wall time of the reference's compiled / external steps of `_build_sppmi` on a stream: sort(1) of the pair lines,
    _parallel_build_sppmi, sort(1) of its output, _chunking_into_bins -- what stock buffalo spends after its Python loop has written
    the pair lines (that loop, stream.py:257-267, is NOT timed: here the lines are written by numpy, in arbitrary order)."""
    import time
    indptr = np.asarray(indptr, dtype=np.int64)
    items = np.asarray(items, dtype=np.int64) + 1
    user = np.repeat(np.arange(indptr.shape[0]), np.diff(np.concatenate([[0], indptr])))
    parts = []
    for w in range(1, int(windows) + 1):
        same = user[:-w] == user[w:]
        a, b = items[:-w][same], items[w:][same]
        parts.append(np.stack([a, b], 1))
        parts.append(np.stack([b, a], 1))
    pairs = np.concatenate(parts) if parts else np.zeros((0, 2), np.int64)
    total_lines = int(pairs.shape[0])
    out = {"total_lines": total_lines, "workers": int(num_workers)}
    with tempfile.TemporaryDirectory() as d:
        src, dst = os.path.join(d, "pairs.txt"), os.path.join(d, "sppmi.txt")
        np.savetxt(src, pairs, fmt="%d %d")
        t0 = time.perf_counter()
        _psort_first_field(src, 1)
        t1 = time.perf_counter()
        nnz = int(lib().ref_parallel_build_sppmi(src.encode(), dst.encode(), total_lines, int(num_items), int(k), int(num_workers)))
        t2 = time.perf_counter()
        _psort_first_field(dst, 1)
        t3 = time.perf_counter()
        if nnz:
            chunks = 2 * max(1, min(int(num_workers), 20))
            lib().ref_chunking_into_bins(dst.encode(), d.encode(), nnz, chunks, 0, max(1, min(int(num_workers), 20)))
        t4 = time.perf_counter()
    out.update(nnz=nnz, sort_lines_s=t1 - t0, build_s=t2 - t1, sort_output_s=t3 - t2, chunk_s=t4 - t3, total_s=t4 - t0)
    return out


def canonical_rows(group):
    """(indptr, key, val) with every row's entries ordered by (key, val): the order-free form two SPPMI groups are compared in."""
    indptr, key, val = group["indptr"], group["key"], group["val"]
    rows = np.repeat(np.arange(indptr.shape[0]), np.diff(np.concatenate([[0], indptr])))
    order = np.lexsort((val, key, rows))
    return {"indptr": indptr.copy(), "key": key[order], "val": val[order]}
