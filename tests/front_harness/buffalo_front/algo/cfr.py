"""CFR (CoFactor) front: what stock buffalo's `buffalo/algo/cfr.py` asks of `CyCFR`, reduced to the training loop.  Written
against the call traces the reference's own class produces (tests/golden/make_front_traces.py, cases cfr_*) and checked
against them call by call in tests/test_front_trace_cpu.py."""
import numpy as np

from buffalo_amd.backend import CyCFR
from ..data import BufferedDataMatrix
from .base import Algo, Evaluable
from .options import CFROption

# (attribute, embedding name handed to the backend, rows from which header field, columns)
_EMBEDDINGS = (("U", "user", "num_users", None), ("I", "item", "num_items", None), ("C", "context", "num_items", None),
               ("Ib", "item_bias", "num_items", 1), ("Cb", "context_bias", "num_items", 1))


class CFR(Algo, CFROption, Evaluable):
    def __init__(self, opt_path=None, *args, **kwargs):
        Algo.__init__(self)
        CFROption.__init__(self, *args, **kwargs)
        Evaluable.__init__(self)
        self._open("CFR", CyCFR, opt_path, kwargs, ["stream"])     # cfr.py:52-62: a Stream in matrix layout
        self.is_initialized = False

    def normalize(self, group="item"):
        self._normalize_once(group, {"user": ("U", "_nrz_U"), "item": ("I", "_nrz_I"), "context": ("C", "_nrz_C")})

    def initialize(self):  # cfr.py:85-103: N(0, 1/d^2) embeddings (signed, unlike ALS / BPRMF), each bound to the backend by name
        super().initialize()
        assert self.data, "Data is not set"
        header, d = self.data.get_header(), self.opt.d
        for attr, name, rows, cols in _EMBEDDINGS:
            F = np.random.normal(scale=1.0 / (d ** 2), size=(header[rows], cols or d)).astype(np.float32)
            setattr(self, attr, F)
            self.obj.set_embedding(F, name.encode("utf8"))
        self.P, self.Q = self.U, self.I
        self.is_initialized = True

    def _sweep(self, buf, group):
        """One of the three block updates of an epoch: which Gramian is refreshed first, which groups size the row ranges."""
        if group == "user":
            self.obj.precompute(b"item")
            sized_by = ["rowwise"]
        elif group == "item":
            self.obj.precompute(b"user")
            sized_by = ["colwise", "sppmi"]
        else:
            sized_by = ["sppmi"]
        err = 0
        for start_x, next_x in buf.fetch_batch_range(sized_by):
            if group == "user":
                err += self.obj.partial_update_user(start_x, next_x, *buf.get_specific_chunk("rowwise", start_x, next_x))
            elif group == "item":
                err += self.obj.partial_update_item(start_x, next_x, *buf.get_specific_chunk("colwise", start_x, next_x),
                                                    *buf.get_specific_chunk("sppmi", start_x, next_x))
            else:
                err += self.obj.partial_update_context(start_x, next_x, *buf.get_specific_chunk("sppmi", start_x, next_x))
        return err

    def compute_scale(self):  # cfr.py:184-191: what the summed loss is divided by
        header = self.data.get_header()
        vals = self.data.get_group("rowwise")["val"]
        vsum = 0.0
        for beg in range(0, header["num_nnz"], 100000):      # data/base.py:73-80: summed in slices (the partial sums are float32,
            vsum += np.sum(vals[beg:beg + 100000])            # and numpy keeps the running sum float32: the scale is too)
        sppmi_nnz = int(self.data.get_group("sppmi")["key"].shape[0])
        return self.opt.l * (self.opt.alpha * vsum + header["num_users"] * header["num_items"]) + sppmi_nnz

    def train(self, training_callback=None):
        assert self.is_initialized, "embedding matrix is not initialized"
        buf = BufferedDataMatrix()
        buf.initialize(self.data, with_sppmi=True)
        scale = self.compute_scale()
        loss = self._epochs(lambda _: sum(self._sweep(buf, g) for g in ("user", "item", "context")) / scale, training_callback, prefix="vali_")
        return self._result(0.0 if loss is None else loss, prefix="vali_")

    def _get_data(self):
        return super()._get_data() + [("opt", self.opt), ("I", self.I), ("U", self.U), ("C", self.C)]
