"""EALS front: what stock buffalo's `buffalo/algo/eals.py` asks of `CyEALS`, reduced to the training loop.  Written against the
call trace the reference's own class produces (tests/golden/make_front_traces.py, case eals) and checked against it call by
call in tests/test_front_trace_cpu.py."""
import json

import numpy as np

from buffalo_amd.backend import CyEALS
from ..data import Data
from .base import Algo, Evaluable, get_logger
from .options import EALSOption

_AXIS = {"rowwise": 0, "colwise": 1}


class EALS(Algo, EALSOption, Evaluable):
    def __init__(self, opt_path=None, *args, **kwargs):
        Algo.__init__(self)
        EALSOption.__init__(self, *args, **kwargs)
        Evaluable.__init__(self)
        self.logger = get_logger("EALS")
        self.opt, self.opt_path = self.get_option(EALSOption().get_default_option() if opt_path is None else opt_path)
        self.obj = CyEALS()
        assert self.obj.init(bytes(self.opt_path, "utf-8")), "cannot parse option file: %s" % opt_path
        data = kwargs.get("data")
        self.data = data if isinstance(data, Data) else None
        self.logger.info("eALS(%s)" % json.dumps(self.opt, indent=2))
        if self.data:
            assert self.data.data_type in ["matrix"]

    def normalize(self, group="item"):
        if group == "item" and not self.opt._nrz_Q:
            self.Q, self.opt._nrz_Q = self._normalize(self.Q), True
        elif group == "user" and not self.opt._nrz_P:
            self.P, self.opt._nrz_P = self._normalize(self.P), True

    def initialize(self):
        super().initialize()
        self.init_factors()

    def _group(self, name):
        g = self.data.get_group(name)
        return g["indptr"][:], g["key"][:], g["val"][:]

    def negative_weights(self):
        """eals.py:104-112: c0 times the items' popularity (entries per item over the largest) to the power `exponent`,
        normalised to sum 1 -- float32 throughout, as the reference computes it."""
        indptr = self._group("colwise")[0]
        pop = np.array([indptr[i] - (indptr[i - 1] if i else 0) for i in range(len(indptr))], dtype="float32")
        pop /= max(pop)
        powered = pop ** self.opt.get("exponent", 0.0)
        return self.opt.get("c0", 1.0) * powered / sum(powered)

    def init_factors(self):  # eals.py:74-87: signed N(0, 1/d^2) factors, no padding (vdim = d)
        assert self.data, "Data is not set"
        header, d = self.data.get_header(), self.opt.d
        self.vdim, self._nnz = d, header["num_nnz"]
        self.P = np.random.normal(scale=1.0 / (d ** 2), size=(header["num_users"], d)).astype("float32")
        self.Q = np.random.normal(scale=1.0 / (d ** 2), size=(header["num_items"], d)).astype("float32")
        self.C = self.negative_weights()
        self.obj.initialize_model(self.P, self.Q, self.C)

    def train(self, training_callback=None):
        best_loss, loss, self.validation_result = float("inf"), None, {}
        for name in ("rowwise", "colwise"):
            indptr, keys, _ = self._group(name)
            self.obj.precompute_cache(self._nnz, indptr, keys, _AXIS[name])
        for i in range(self.opt.num_iters):
            for name in ("rowwise", "colwise"):
                assert self.obj.update(*self._group(name), _AXIS[name])
            loss, total = self.obj.estimate_loss(self._nnz, *self._group("rowwise"), _AXIS["rowwise"])
            metrics = {"train_loss": loss}
            if self.opt.validation and self.opt.evaluation_on_learning and self.periodical(self.opt.evaluation_period, i):
                self.validation_result = self.get_validation_results()
                metrics.update({"val_%s" % k: v for k, v in self.validation_result.items()})
                if callable(training_callback):
                    training_callback(i, metrics)
            best_loss = self.save_best_only(loss, best_loss, i)
            if self.early_stopping(loss):
                break
        ret = {"train_loss": loss}
        ret.update({"val_%s" % k: v for k, v in self.validation_result.items()})
        return ret

    def _get_data(self):
        return super()._get_data() + [("opt", self.opt), ("Q", self.Q), ("P", self.P)]
