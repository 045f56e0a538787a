"""In-memory data objects feeding the training core (stand-ins for buffalo.data's MatrixMarket / Stream / BufferedDataMatrix).

The reference stages its inputs into an HDF5 file through text files and a C++ sorter (/root/reference/buffalo/data/{base,mm,stream}.py,
fileio.hpp).  Here the same GROUPS are produced in memory -- rowwise / colwise {indptr int64[rows] END offsets, key int32, val float32},
`vali` {row, col, val}, `sppmi` -- with the sort + compression (and the SPPMI build) done on the device (`buffalo_amd.ingest`).
What the loaders store is held, array by array, to databases built by the reference's own data package run end to end
(tests/golden/make_data_vectors.py, tests/test_data_loaders_ref.py): same records in the same order (entries sharing a cell stay
apart), the same np.random draws for `sample` validation, the reference's validation layout and its quirks (cited where they are).
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


def _read_ids(src, n, what, first=0):
    if src is None or (isinstance(src, str) and src == ""):
        return [str(i) for i in range(first, n + first)]      # mm.py:143, stream.py:145: the reference names them "1".."n"
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

    def _finish(self, num_users, num_items, rows, cols, vals, vali, num_nnz=None, colwise=True):
        """Both orientations from the training records IN THE ORDER GIVEN (the reference's sort is stable on (row, col), so records
        that share a cell stay apart and in input order: fileio.hpp:327-341), the `vali` group, the header."""
        rows = np.asarray(rows, dtype=np.int64)
        cols = np.asarray(cols, dtype=np.int64)
        vals = np.asarray(vals, dtype=np.float32)
        if num_nnz is not None and rows.shape[0] > num_nnz:
            # the header count is fixed before the split (base.py:224-226) and the sorter keeps the first `total_lines` records
            # of the working file (fileio.hpp:313-323): what does not fit is dropped
            rows, cols, vals = rows[:num_nnz], cols[:num_nnz], vals[:num_nnz]
        if colwise:
            self.groups["rowwise"] = _group(num_users, num_items, rows, cols, vals)
            self.groups["colwise"] = _group(num_items, num_users, cols, rows, vals)
        else:
            # internal_data_type "stream" (stream.py:160-163, sort_key -1): the events stay in their order, so the row-wise group is the
            # records as they come plus END offsets; the column-wise group is allocated (base.py:185-192) and never filled
            self.groups["rowwise"] = {"indptr": np.cumsum(np.bincount(rows, minlength=num_users)).astype(np.int64),
                                      "key": cols.astype(np.int32), "val": vals}
            self.groups["colwise"] = {"indptr": np.zeros(num_items, np.int64), "key": np.zeros(rows.shape[0], np.int32),
                                      "val": np.zeros(rows.shape[0], np.float32)}
        if vali is not None and len(vali[0]):
            vr, vc, vv = (np.asarray(x) for x in vali)
            # base.py:241-253: rows and columns in the order the samples were met, the VALUES in the order of a CSR built from
            # them (scipy sorts by row, then column) -- the two orders differ whenever the samples are not already sorted
            vsorted = scipy.sparse.csr_matrix((vv.astype(np.float32), (vr, vc)), (num_users, num_items)).data
            self.groups["vali"] = {"row": vr.astype(np.int32), "col": vc.astype(np.int32), "val": vsorted.astype(np.float32)}
        self.header = {"num_nnz": int(rows.shape[0]), "num_users": int(num_users), "num_items": int(num_items), "completed": 1}
        return self

    def _sample_size(self, num_nnz):
        """base.py:220-226: how many entries the `sample` method holds out, and which (0-based positions; the last one is never
        drawn, the reference's reader cannot split it off)."""
        v = self.opt.data.validation
        if not v or v.get("name") != "sample":
            return None
        sz = min(int(v.get("max_samples", 500)), int(num_nnz * v.get("p", 0.01)))
        return np.random.choice(num_nnz - 1, sz, replace=False)


class MatrixMarket(Data):
    name = "MatrixMarket"

    def _records(self):
        """(U, I, rows, cols, vals) in FILE order.  A matrix handed over in memory goes through scipy's writer first, as in the
        reference (mm.py:63-80), which lists a CSR matrix row by row."""
        main = self.opt.input.main
        if not isinstance(main, str):
            if isinstance(main, np.ndarray) and main.ndim == 2:
                main = scipy.sparse.csr_matrix(main)
            if not scipy.sparse.issparse(main):
                raise RuntimeError("Unexpected data type for MatrixMarketOption.input.main field: %s" % type(main))
            coo = main.tocoo()
            return coo.shape[0], coo.shape[1], coo.row.astype(np.int64), coo.col.astype(np.int64), coo.data.astype(np.float32)
        with open(main) as fin:
            body = [l for l in fin if not l.lstrip().startswith("%")]
        U, I, _ = (int(x) for x in body[0].split())
        rec = np.array([l.split()[:3] for l in body[1:] if l.strip()], dtype=np.float64).reshape(-1, 3)
        return U, I, rec[:, 0].astype(np.int64) - 1, rec[:, 1].astype(np.int64) - 1, rec[:, 2].astype(np.float32)

    def create(self):
        """mm.py:236-279: the coordinate lines in file order; `sample` validation takes drawn LINES out (mm.py:167-234); what is
        left is sorted (stably) into both orientations."""
        U, I, rows, cols, vals = self._records()
        self.userids = _read_ids(self.opt.input.uid, U, "uid", first=1)
        self.itemids = _read_ids(self.opt.input.iid, I, "iid", first=1)
        picked = self._sample_size(rows.shape[0])
        vali = None
        keep = np.ones(rows.shape[0], dtype=bool)
        if picked is not None and len(picked):
            idx = np.sort(picked)                         # mm.py:186: taken out in the order the reader meets them
            keep[idx] = False
            vali = (rows[idx], cols[idx], vals[idx])
        return self._finish(U, I, rows[keep], cols[keep], vals[keep], vali)


def _sppmi_group(indptr, items, num_items, windows, k):
    """The `sppmi` group of a stream (stream.py:169-195), built on the device."""
    from buffalo_amd.ingest import build_sppmi
    g = build_sppmi(indptr, items, num_items, windows, k)
    return {"indptr": g["indptr"], "key": g["key"], "val": g["val"]}


def _counted(seq):
    """collections.Counter(seq).items(): distinct entries in order of first appearance, with their counts."""
    counts = {}
    for x in seq:
        counts[x] = counts.get(x, 0) + 1
    return counts.items()


class Stream(Data):
    name = "Stream"
    data_type = "stream"           # stream.py:79: what CFR asks for, whatever the internal layout

    def create(self):
        """stream.py:273-317.  Every line is one user's item sequence.  `newest` validation holds out the last n events of a
        sequence (never its only one; a held-out item counts once, stream.py:222-230), `sample` validation the events at drawn
        positions of the concatenated sequences (:231-245).  internal_data_type "matrix": a user's remaining events become
        (item, count) records in order of first appearance; "stream": one record per event, order kept, row-wise only.  With
        data.sppmi = {windows, k} the remaining events, in their order, also feed the `sppmi` group (:257-267, 169-195)."""
        with open(self.opt.input.main) as fin:
            lines = [l.split() for l in fin]
        U = len(lines)
        self.userids = _read_ids(self.opt.input.uid, U, "uid", first=1)
        iid = self.opt.input.iid
        if iid is None or iid == "":
            names = sorted({w for l in lines for w in l})       # the reference numbers them in set order (stream.py:122-123): arbitrary
        else:
            names = _read_ids(iid, len(open(iid).readlines()) if isinstance(iid, str) else len(iid), "iid")
        self.itemids = names
        index = {w: i for i, w in enumerate(names)}
        v = self.opt.data.validation
        as_matrix = self.opt.data.internal_data_type == "matrix"
        vali_n = int(v.get("n", 0)) if v and v.get("name") == "newest" else 0
        kept = [len(l) - min(vali_n, max(len(l) - 1, 0)) for l in lines]
        # stream.py:100-120: the header count is taken before any `sample` draw -- distinct items per user in matrix layout
        counted = sum(len(set(l[:k])) if as_matrix else k for l, k in zip(lines, kept))
        picked = self._sample_size(counted)
        positions = set() if picked is None else set(int(p) for p in picked)
        rows, cols, vals, vr, vc, vv = [], [], [], [], [], []
        seq_end, seq_items = [], []
        at = 0
        for u, (seq, k) in enumerate(zip(lines, kept)):
            held = [index[w] for w, _ in _counted(seq[k:])]
            train = []
            for pos, w in enumerate(seq[:k]):
                (held if at + pos in positions else train).append(index[w])
            at += k
            seq_items.extend(train)
            seq_end.append(len(seq_items))
            for c, cnt in (_counted(train) if as_matrix else ((c, 1) for c in train)):
                rows.append(u), cols.append(c), vals.append(float(cnt))
            for c, cnt in _counted(held):
                vr.append(u), vc.append(c), vv.append(float(cnt))
        vali = (vr, vc, vv) if vr else None
        self._finish(U, len(names), rows, cols, vals, vali, num_nnz=counted - (0 if picked is None else len(picked)), colwise=as_matrix)
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
