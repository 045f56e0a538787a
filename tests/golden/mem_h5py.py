"""An in-memory stand-in for the part of h5py the reference's data package touches (TEST INFRASTRUCTURE).

h5py is not installed in this image, and the HDF5 container is outside the scope table; what matters is what the reference's own
`buffalo/data/{base,mm,stream}.py` -- imported unmodified from /root/reference -- WRITE into it.  This module gives them
`File` / groups / datasets / `attrs` / `string_dtype` over numpy arrays held in a per-path registry, so `MatrixMarket.create()` and
`Stream.create()` run here end to end (with the reference's own compiled `fileio.hpp` behind `buffalo.data.fileio`, oracle/_ref)
and the groups they build can be read back and committed as golden vectors (tests/golden/make_data_vectors.py).

Surface (from grep over the reference: data/base.py:50-71,181-237, mm.py:141-151,267-269, stream.py:183-188,306-308):
`File(path, mode)`, `.close()`, `.attrs`, `create_group`, `create_dataset(name, shape, dtype=, maxshape=, chunks=)`, `[]`, `in`,
`.keys()`, dataset slicing / assignment / iteration / `.shape`, `string_dtype("utf-8", length=n)`.
"""
import os

import numpy as np

_FILES = {}   # absolute path -> (items, attrs) of the root group


def string_dtype(encoding="utf-8", length=None):
    """Fixed-length byte strings, like h5py's (whose elements also come back as bytes and are .decode()d by the reference)."""
    assert length is not None, "only fixed-length strings are used by the reference"
    return np.dtype("S%d" % int(length))


def special_dtype(**kwargs):
    return np.dtype(object)


class Dataset:
    def __init__(self, shape, dtype, chunks=None):
        self._a = np.zeros(shape, dtype=np.dtype(dtype))
        self.chunks = tuple(chunks) if chunks else tuple(max(1, min(n, 1 << 16)) for n in self._a.shape)   # prepro.py:53 reads chunks[0]

    def _coerce(self, value):
        if self._a.dtype.kind == "S":   # h5py encodes str with the dtype's encoding; numpy's S refuses non-ASCII str
            if isinstance(value, str):
                return value.encode("utf-8")
            if isinstance(value, (list, tuple)):
                return [v.encode("utf-8") if isinstance(v, str) else v for v in value]
            if isinstance(value, np.ndarray) and value.dtype.kind == "U":
                return np.char.encode(value, "utf-8")
        return value

    def __setitem__(self, key, value):
        try:
            self._a[key] = self._coerce(value)
        except ValueError as e:            # h5py reports a selection / shape mismatch as TypeError ("Can't broadcast ...")
            raise TypeError(str(e)) from e

    def __getitem__(self, key):
        return self._a[key]

    def __len__(self):
        return len(self._a)

    def __iter__(self):
        return iter(self._a)

    def __array__(self, dtype=None, copy=None):
        return self._a if dtype is None else self._a.astype(dtype)

    @property
    def shape(self):
        return self._a.shape

    @property
    def dtype(self):
        return self._a.dtype


class Group:
    def __init__(self, items=None, attrs=None):
        self._items = {} if items is None else items
        self.attrs = {} if attrs is None else attrs

    def create_group(self, name):
        assert name not in self._items, "group %s exists" % name
        g = self._items[name] = Group()
        return g

    def create_dataset(self, name, shape=None, dtype=None, data=None, maxshape=None, chunks=None, **kwargs):
        assert name not in self._items, "dataset %s exists" % name
        if data is not None:
            data = np.asarray(data)
            d = Dataset(data.shape, dtype or data.dtype, chunks)
            d[...] = data
        else:
            d = Dataset(shape, dtype, chunks)
        self._items[name] = d
        return d

    def __getitem__(self, name):
        return self._items[name]

    def __contains__(self, name):
        return name in self._items

    def keys(self):
        return self._items.keys()

    def __iter__(self):
        return iter(self._items)


class File(Group):
    def __init__(self, path, mode="r", **kwargs):
        key = os.path.abspath(path)
        if mode == "w":
            _FILES[key] = ({}, {})
            open(path, "wb").close()          # the reference tests os.path.isfile(path) / os.remove(path) on it
        elif key not in _FILES:
            raise OSError("Unable to open file (no such in-memory file: %s)" % path)
        items, attrs = _FILES[key]
        super().__init__(items, attrs)
        self.filename = path

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def tree(group):
    """{name: ndarray | nested dict} + {"@attrs": {...}}: what the reference wrote, as plain numpy."""
    out = {"@attrs": dict(group.attrs)}
    for name in group.keys():
        item = group[name]
        out[name] = tree(item) if isinstance(item, Group) else np.array(item[...])
    return out
