"""In-memory data objects feeding the training core (shim plumbing).

The reference stages MatrixMarket / Stream inputs into an HDF5 file through a C++ sorter
(/root/reference/buffalo/data/{base,mm,stream}.py, fileio.hpp); h5py is absent here and ingestion is
outside the hot path, so the same *layout* is produced directly in numpy:
rowwise / colwise groups of {indptr int64[rows] END offsets, key int32 sorted per row, val float32}
plus a `vali` group -- exactly what `BufferedDataMatrix` and the algo classes consume.
"""
import bisect

import numpy as np
import scipy.io
import scipy.sparse

from ..misc import InputOptions, Option


class DataOption(InputOptions):
    def is_valid_option(self, opt) -> bool:
        default = self.get_default_option()
        for section in ("type", "input", "data"):
            if section not in opt:
                raise RuntimeError("{} not exists on Option".format(section))
        for k in default["data"]:
            opt["data"].setdefault(k, default["data"][k])
        return True


class MatrixMarketOptions(DataOption):
    def get_default_option(self) -> Option:  # mm.py:15-37
        return Option({
            "type": "matrix_market",
            "input": {"main": "", "uid": "", "iid": ""},
            "data": {"internal_data_type": "matrix",
                     "validation": {"name": "sample", "p": 0.01, "max_samples": 500},
                     "batch_mb": 1024, "use_cache": False, "tmp_dir": "/tmp/", "path": "./mm.h5py",
                     "disk_based": False},
        })

    def is_valid_option(self, opt) -> bool:  # mm.py:39-55
        super().is_valid_option(opt)
        if opt["type"] != "matrix_market":
            raise RuntimeError("Invalid data type: %s" % opt["type"])
        if opt["data"]["internal_data_type"] != "matrix":
            raise RuntimeError("MatrixMarket only support internal data type(matrix)")
        return True


class StreamOptions(DataOption):
    def get_default_option(self) -> Option:  # stream.py:38-65
        return Option({
            "type": "stream",
            "input": {"main": "", "uid": "", "iid": ""},
            "data": {"validation": {"name": "newest", "p": 0.01, "n": 1, "max_samples": 500},
                     "sppmi": {}, "batch_mb": 1024, "use_cache": False, "tmp_dir": "/tmp/",
                     "path": "./stream.h5py", "internal_data_type": "matrix", "disk_based": False},
        })

    def is_valid_option(self, opt) -> bool:
        super().is_valid_option(opt)
        if opt["type"] != "stream":
            raise RuntimeError("Invalid data type: %s" % opt["type"])
        return True


def _group(num_rows, num_cols, rows, cols, vals):
    """(row, col)-sorted CSR group in the reference layout (fileio.hpp:263-420), built on the device."""
    from buffalo_amd.ingest import coo_to_csr
    return coo_to_csr(rows, cols, vals, num_rows, num_cols)


def _read_ids(src, n, what):
    if src is None or (isinstance(src, str) and src == ""):
        return [str(i) for i in range(n)]
    if isinstance(src, str):
        with open(src) as fin:
            ids = [l.rstrip("\n") for l in fin]
    else:
        ids = [str(x) for x in src]
    if len(ids) != n:
        raise RuntimeError("%s list has %d entries, expected %d" % (what, len(ids), n))
    return ids


class Data:
    """Base of MatrixMarket / Stream: holds the groups + header + id lists (data/base.py:15-208)."""
    data_type = "matrix"
    name = "Data"

    def __init__(self, opt, *args, **kwargs):
        self.opt = Option(opt)
        self.groups, self.header, self.userids, self.itemids = {}, None, [], []

    # -- reference surface used by the algo classes --------------------------------------------
    def get_header(self):
        return self.header

    def get_group(self, name):
        return self.groups[name]

    def has_group(self, name):
        return name in self.groups

    def show_info(self):
        h = self.header
        vali = self.groups["vali"]["row"].shape[0] if "vali" in self.groups else 0
        return "{} Header({}, {}, {}) Validation({} samples)".format(self.name, h["num_users"], h["num_items"],
                                                                   h["num_nnz"], vali)

    def get(self, index, axis="rowwise"):
        g = self.groups[axis]
        beg = 0 if index == 0 else int(g["indptr"][index - 1])
        end = int(g["indptr"][index])
        return g["key"][beg:end], g["val"][beg:end]

    def close(self):
        pass

    def _finish(self, num_users, num_items, rows, cols, vals, vali):
        rows = np.asarray(rows, dtype=np.int64)
        cols = np.asarray(cols, dtype=np.int64)
        vals = np.asarray(vals, dtype=np.float32)
        self.groups["rowwise"] = _group(num_users, num_items, rows, cols, vals)
        self.groups["colwise"] = _group(num_items, num_users, cols, rows, vals)
        if vali is not None and len(vali[0]):
            self.groups["vali"] = {"row": np.asarray(vali[0], dtype=np.int32), "col": np.asarray(vali[1], dtype=np.int32),
                                   "val": np.asarray(vali[2], dtype=np.float32)}
        self.header = {"num_nnz": int(rows.shape[0]), "num_users": int(num_users), "num_items": int(num_items),
                       "completed": 1}
        return self


class MatrixMarket(Data):
    name = "MatrixMarket"

    def create(self):
        """mm.py:236-279: read, hold out validation samples (`sample`: a random p fraction capped at
        max_samples, mm.py:167-234), build both orientations."""
        main = self.opt.input.main
        if isinstance(main, str):
            M = scipy.io.mmread(main)
        elif scipy.sparse.issparse(main):
            M = main
        else:
            M = scipy.sparse.csr_matrix(np.asarray(main))
        M = scipy.sparse.coo_matrix(M)
        M.sum_duplicates()
        U, I = M.shape
        self.userids = _read_ids(self.opt.input.uid, U, "uid")
        self.itemids = _read_ids(self.opt.input.iid, I, "iid")
        rows, cols, vals = M.row, M.col, M.data.astype(np.float32)
        keep = np.ones(rows.shape[0], dtype=bool)
        vali = None
        v = self.opt.data.validation
        if v and v.get("name") == "sample" and rows.shape[0]:
            n = min(int(rows.shape[0] * v.get("p", 0.01)), int(v.get("max_samples", 500)))
            if n > 0:
                idx = np.random.choice(rows.shape[0], size=n, replace=False)
                keep[idx] = False
                vali = (rows[idx], cols[idx], vals[idx])
        return self._finish(U, I, rows[keep], cols[keep], vals[keep], vali)


def _sppmi_group(indptr, items, num_items, windows, k):
    """The `sppmi` group of a stream (stream.py:169-195), built on the device."""
    from buffalo_amd.ingest import build_sppmi
    g = build_sppmi(indptr, items, num_items, windows, k)
    return {"indptr": g["indptr"], "key": g["key"], "val": g["val"]}


class Stream(Data):
    name = "Stream"
    data_type = "stream"           # stream.py:79: what CFR asks for, whatever the internal layout

    def create(self):
        """stream.py:273-317 with internal_data_type "matrix": every line is one user's item
        sequence; counts become values; `newest` validation holds out the last n items.  With data.sppmi = {windows, k} the
        training part of every sequence, in its order, also feeds the `sppmi` group (stream.py:257-267, 169-195)."""
        with open(self.opt.input.main) as fin:
            lines = [l.split() for l in fin]
        U = len(lines)
        self.userids = _read_ids(self.opt.input.uid, U, "uid")
        iid = self.opt.input.iid
        if iid is None or iid == "":
            names = sorted({w for l in lines for w in l})
        else:
            names = _read_ids(iid, len(open(iid).readlines()) if isinstance(iid, str) else len(iid), "iid")
        self.itemids = names
        index = {w: i for i, w in enumerate(names)}
        v = self.opt.data.validation
        vali_n = int(v.get("n", 0)) if v and v.get("name") == "newest" else 0
        rows, cols, vals, vr, vc, vv = [], [], [], [], [], []
        seq_end, seq_items = [], []
        for u, seq in enumerate(lines):
            ids = [index[w] for w in seq if w in index]
            k = min(vali_n, max(len(ids) - 1, 0))
            train, held = ids[:len(ids) - k], ids[len(ids) - k:]
            seq_items.extend(train)
            seq_end.append(len(seq_items))
            for c, cnt in zip(*np.unique(train, return_counts=True)) if train else ():
                rows.append(u), cols.append(int(c)), vals.append(float(cnt))
            for c, cnt in zip(*np.unique(held, return_counts=True)) if held else ():
                vr.append(u), vc.append(int(c)), vv.append(float(cnt))
        # every held-out entry is a validation sample (the reference holds out for all users and keeps them all,
        # stream.py:100-118, 222-230: vali_limit is the sum over the users) -- nothing leaves train without entering vali
        vali = (vr, vc, vv) if vr else None
        self._finish(U, len(names), rows, cols, vals, vali)
        sp = self.opt.data.get("sppmi")
        if sp:
            self.groups["sppmi"] = _sppmi_group(np.asarray(seq_end, dtype=np.int64), np.asarray(seq_items, dtype=np.int32), len(names),
                                                int(sp["windows"]), int(sp["k"]))
            self.header["sppmi_nnz"] = int(self.groups["sppmi"]["key"].shape[0])
        return self


def load(opt):
    """buffalo.data.load (data/__init__.py): construct from an option dict / Option."""
    opt = Option(opt) if not isinstance(opt, Option) else opt
    if opt["type"] == "matrix_market":
        MatrixMarketOptions().is_valid_option(opt)
        return MatrixMarket(opt)
    if opt["type"] == "stream":
        StreamOptions().is_valid_option(opt)
        return Stream(opt)
    raise RuntimeError("Unexpected data type: %s" % opt["type"])


class BufferedDataMatrix:
    """Chunk feeder with the reference's exact semantics (buffered_data.py:27-172): chunks are
    row-aligned, bounded by batch_mb (16 B per nnz, half the budget per group), `indptr` is the full
    matrix' end offsets and keys/vals hold the current chunk only.  Includes Q-24 (a single trailing
    row is never fed in multi-chunk mode; in single-chunk mode nothing is re-copied after epoch 1)."""

    def __init__(self):
        self.group = "rowwise"
        self.major = {"rowwise": {}, "colwise": {}}

    # -- CFR's feeding (buffered_data.py:122-158): row ranges sized by the SUM of several groups' entries, chunks cut on demand
    def fetch_batch_range(self, groups):
        """Yield (start_x, next_x) ranges such that the entries of all `groups` inside a range fit batch_mb at 8 bytes each.
        The last row on its own is never yielded (the reference stops once next_x + 1 reaches the row count, like Q-24)."""
        total = sum(np.asarray(self.data.get_group(G)["indptr"], dtype=np.int64) for G in groups)
        budget = max(int(self.data.opt.data.batch_mb * 1024 * 1024 / 8.), 64)
        rows, start = len(total), 0
        while True:
            beg = int(total[start - 1]) if start else 0
            nxt = bisect.bisect_left(total, beg + budget)
            if nxt == start:
                raise RuntimeError("Need more memory to load the data, cannot load data with buffer size %d that should be at "
                                   "least %d. Increase batch_mb value to deal with this." % (budget, total[nxt] - beg))
            yield start, nxt
            if nxt + 1 >= rows:
                return
            start = nxt

    def get_specific_chunk(self, group, start_x, next_x):
        """(full END-offset indptr of `group`, its keys and values for rows [start_x, next_x))."""
        g = self.data.get_group(group)
        indptr = self.major[group]["indptr"] if group in self.major and self.major[group] else g["indptr"]
        beg = int(indptr[start_x - 1]) if start_x else 0
        end = int(indptr[next_x - 1])
        return indptr, g["key"][beg:end], g["val"][beg:end]

    def get_indptrs(self):
        return (self.major["rowwise"]["indptr"], self.major["colwise"]["indptr"], self.major["rowwise"]["limit"])

    def initialize(self, data, with_sppmi=False):
        self.data = data
        limit = max(int((self.data.opt.data.batch_mb * 1024 * 1024) / 16.), 64)
        need = 0
        if with_sppmi:
            self.major["sppmi"] = {}
        for G in ("rowwise", "colwise") + (("sppmi",) if with_sppmi else ()):
            lim = int(limit / 2)
            g, header = data.get_group(G), data.get_header()
            m = self.major[G] = {"index": 0, "limit": lim, "start_x": 0, "next_x": 0,
                                 "max_x": header["num_users"] if G == "rowwise" else header["num_items"],
                                 "indptr": g["indptr"]}
            need = max(need, int(np.max(np.diff(m["indptr"]))) if len(m["indptr"]) > 1 else 0)  # buffered_data.py:65-66
            m["keys"] = np.zeros(lim, dtype=np.int32)
            m["vals"] = np.zeros(lim, dtype=np.float32)
        if need > int(limit / 2):
            for G in ("rowwise", "colwise"):
                m = self.major[G]
                m["limit"] = need + 1
                m["keys"] = np.zeros(need + 1, dtype=np.int32)
                m["vals"] = np.zeros(need + 1, dtype=np.float32)

    def fetch_batch(self):
        m = self.major[self.group]
        flushed = False
        while True:
            if m["start_x"] == 0 and m["next_x"] + 1 >= m["max_x"]:
                if not flushed:
                    m["sz"] = m["indptr"][-1]
                    yield m["indptr"][-1]
                return
            if m["next_x"] + 1 >= m["max_x"]:
                m["start_x"], m["next_x"] = 0, 0
                return
            m["start_x"] = m["next_x"]
            group = self.data.get_group(self.group)
            beg = 0 if m["start_x"] == 0 else m["indptr"][m["start_x"] - 1]
            where = bisect.bisect_left(m["indptr"], beg + m["limit"])
            if where == m["start_x"]:
                raise RuntimeError("Need more memory to load the data, cannot load data with buffer size %d that "
                                   "should be at least %d. Increase batch_mb value to deal with this."
                                   % (m["limit"], m["indptr"][where] - beg))
            end = m["indptr"][where - 1]
            m["next_x"] = where
            size = end - beg
            m["keys"][:size] = group["key"][beg:end]
            m["vals"][:size] = group["val"][beg:end]
            if m["next_x"] + 1 >= m["max_x"]:
                flushed = True
            m["sz"] = size
            yield size

    def set_group(self, group):
        assert group in ("rowwise", "colwise"), "Unexpected group: {}".format(group)
        self.group = group

    def get(self):
        m = self.major[self.group]
        return [m[k] for k in ("start_x", "next_x", "indptr", "keys", "vals")]
