"""Model files byte-compatible with buffalo's `Serializable` (/root/reference/buffalo/algo/base.py:271-318).

Format: u64 object count, then per object `u64 len | utf-8 name | u64 len | pickle (protocol 4)`.  The
objects are `_idmanager` and `opt` -- instances of `buffalo.misc._aux.Option` (a dict subclass) -- and the
numpy factor matrices.  A pickle names classes by module path, so

* when WRITING, our `Option` is emitted under the reference's path `buffalo.misc._aux.Option`
  (stock buffalo then loads the file with its own class);
* when READING, that path resolves to our `Option` if the buffalo package itself is not importable.

The two classes have the same pickling behaviour (`__getstate__` / `__setstate__` over `vars()`, dict items), and a
file written here is byte-identical to the one stock buffalo writes for the same content
(`tests/golden/model_ref.bin`, produced with the reference's own class by `tests/golden/make_model_fixture.py`)."""
import contextlib
import io
import pickle
import struct
import sys
import threading
import types



class Option(dict):
    """Attribute dict that pickles like `buffalo.misc._aux.Option` (/root/reference/buffalo/misc/_aux.py:16-56): stock
    buffalo's class is a dict subclass that mirrors its items into the instance `__dict__` and pickles that mirror as its
    state, so a dump carries the items twice -- as dict items and as state -- with the state's keys memo-shared with the
    item keys.  Here nothing is mirrored: attributes ARE the items, and `__getstate__` hands pickle a plain dict built from
    the items' own key objects, which yields the very same byte stream (tests/golden/model_ref.bin) whatever instances
    existed before.  Only what model files need lives here: dicts handed to the constructor become Options (as in the
    reference), attribute reads of missing keys give None; the option *validation* of buffalo's front is not product code."""

    def __init__(self, *sources, **items):
        super().__init__()
        for src in sources + (items,):
            for key, value in dict(src).items():
                self[key] = Option(value) if isinstance(value, dict) and not isinstance(value, Option) else value

    def __getattr__(self, name):
        if name.startswith("__"):           # pickle / copy probe for dunder hooks: those are not options
            raise AttributeError(name)
        return self.get(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]

    def __getstate__(self):
        return dict(self)

    def __setstate__(self, state):
        self.update(state)


_REF_MODULE = "buffalo.misc._aux"
_lock = threading.Lock()


@contextlib.contextmanager
def _option_under_reference_path():
    """Make `pickle` see our Option as `buffalo.misc._aux.Option` for the duration of a dump."""
    with _lock:
        real = sys.modules.get(_REF_MODULE)
        if real is not None and getattr(real, "Option", None) is not Option:
            # stock buffalo is importable in this process: convert instead of aliasing
            yield real.Option
            return
        saved = {name: sys.modules.get(name) for name in ("buffalo", "buffalo.misc", _REF_MODULE)}
        old_module = Option.__module__
        try:
            for name in ("buffalo", "buffalo.misc", _REF_MODULE):
                if sys.modules.get(name) is None:
                    sys.modules[name] = types.ModuleType(name)
            sys.modules[_REF_MODULE].Option = Option
            Option.__module__ = _REF_MODULE
            yield Option
        finally:
            Option.__module__ = old_module
            for name, mod in saved.items():
                if mod is None:
                    sys.modules.pop(name, None)
                else:
                    sys.modules[name] = mod


def _convert(obj, cls):
    if isinstance(obj, dict) and (isinstance(obj, Option) or type(obj).__name__ == "Option") and type(obj) is not cls:
        return cls({k: _convert(v, cls) for k, v in obj.items()})
    return obj


def dump_objects(path, data):
    """`Serializable.save` (base.py:275-294) for a list of (name, object)."""
    with _option_under_reference_path() as cls, open(path, "wb") as fout:
        fout.write(struct.pack("Q", len(data)))
        for name, obj in data:
            bname = bytes(name, encoding="utf-8")
            blob = pickle.dumps(_convert(obj, cls), protocol=4)
            fout.write(struct.pack("Q", len(bname)) + bname + struct.pack("Q", len(blob)) + blob)


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if name == "Option" and module in (_REF_MODULE, "buffalo.misc.aux", "buffalo_amd.misc", "buffalo_amd.serialize"):
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                return Option
        return super().find_class(module, name)


def load_objects(path, data_fields=()):
    """`Serializable.load` (base.py:300-311): yields (name, object) for the requested fields."""
    out = []
    with open(path, "rb") as fin:
        (total,) = struct.unpack("Q", fin.read(8))
        for _ in range(total):
            (n,) = struct.unpack("Q", fin.read(8))
            name = fin.read(n).decode("utf8")
            (m,) = struct.unpack("Q", fin.read(8))
            if data_fields and name not in data_fields:
                fin.seek(m, 1)
                continue
            out.append((name, _Unpickler(io.BytesIO(fin.read(m))).load()))
    return out
