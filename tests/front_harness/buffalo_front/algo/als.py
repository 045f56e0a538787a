"""ALS front: what stock buffalo's `buffalo/algo/als.py` asks of `CuALS`, reduced to the training loop and checked call by call
against the trace the reference's own class produces (tests/test_front_trace_cpu.py, cases als_*)."""
import numpy as np

from buffalo_amd.backend import CyALS
from ..data import BufferedDataMatrix
from .base import Algo, Evaluable
from .options import ALSOption

_SIDES = (("rowwise", 0), ("colwise", 1))


class ALS(Algo, ALSOption, Evaluable):
    def __init__(self, opt_path=None, *args, **kwargs):
        Algo.__init__(self)
        ALSOption.__init__(self, *args, **kwargs)
        Evaluable.__init__(self)
        self._open("ALS", CyALS, opt_path, kwargs, ["matrix"], accelerator_only=True)

    def normalize(self, group="item"):
        self._normalize_once(group, {"item": ("Q", "_nrz_Q"), "user": ("P", "_nrz_P")})

    def initialize(self):
        super().initialize()
        self.init_factors()

    def init_factors(self):  # als.py:79-89: |N(0, 1/d^2)| in the backend's padded width, pad columns zero
        assert self.data, "Data is not set"
        header, d = self.data.get_header(), self.opt.d
        self.vdim = self.obj.get_vdim()
        for attr, rows in (("P", header["num_users"]), ("Q", header["num_items"])):
            F = np.abs(np.random.normal(scale=1.0 / (d ** 2), size=(rows, self.vdim)).astype("float32"))
            F[:, d:] = 0.0
            setattr(self, attr, F)
        self.obj.initialize_model(self.P, self.Q)

    def _half_epoch(self, buf, group, axis):
        """One side of an epoch: the other side's Gramian, then every chunk of `group`; returns (squared error, weight)."""
        self.obj.precompute(axis)
        buf.set_group(group)
        nume = deno = 0.0
        for _ in buf.fetch_batch():
            start_x, next_x, indptr, keys, vals = buf.get()
            n, w = self.obj.partial_update(start_x, next_x, indptr, keys, vals, axis)
            nume, deno = nume + n, deno + w
        return nume, deno

    def train(self, training_callback=None):
        self.obj.initialize_model(self.P, self.Q)
        buf = BufferedDataMatrix()
        buf.initialize(self.data)
        self.obj.set_placeholder(*buf.get_indptrs())

        def epoch(_):
            (n1, w1), (n2, w2) = (self._half_epoch(buf, g, a) for g, a in _SIDES)
            return ((n1 + n2) / (w1 + w2 + self.opt.eps)) ** 0.5

        rmse = self._epochs(epoch, training_callback, report="RMSE")
        self.P, self.Q = self.P[:, :self.opt.d], self.Q[:, :self.opt.d]
        return self._result(rmse)

    def _get_data(self):
        return super()._get_data() + [("opt", self.opt), ("Q", self.Q), ("P", self.P)]

    def get_evaluation_metrics(self):
        return ["train_loss", "val_rmse", "val_ndcg", "val_map", "val_accuracy", "val_error"]
