"""ALS front (mirror of /root/reference/buffalo/algo/als.py)."""
import json
import time

import numpy as np

from .. import data as bdata
from buffalo_amd.backend import CyALS
from ..data import BufferedDataMatrix, Data
from .base import Algo, Evaluable, get_logger
from .options import ALSOption


class ALS(Algo, ALSOption, Evaluable):
    def __init__(self, opt_path=None, *args, **kwargs):
        Algo.__init__(self)
        ALSOption.__init__(self, *args, **kwargs)
        Evaluable.__init__(self)
        if opt_path is None:
            opt_path = ALSOption().get_default_option()
        self.logger = get_logger("ALS")
        self.opt, self.opt_path = self.get_option(opt_path)
        if not self.opt.accelerator:
            raise NotImplementedError("buffalo_amd provides the accelerator (MI355X) backend only; "
                                      "set accelerator=True or use kakao/buffalo for the CPU path")
        self.obj = CyALS()
        assert self.obj.init(bytes(self.opt_path, "utf-8")), "cannot parse option file: %s" % opt_path
        self.data = None
        data = kwargs.get("data")
        data_opt = kwargs.get("data_opt", self.opt.get("data_opt"))
        if data_opt:
            self.data = bdata.load(data_opt)
            self.data.create()
        elif isinstance(data, Data):
            self.data = data
        self.logger.info("ALS(%s)" % json.dumps(self.opt, indent=2))
        if self.data:
            assert self.data.data_type in ["matrix"]

    def set_data(self, data):
        assert isinstance(data, Data), "Wrong instance: {}".format(type(data))
        self.data = data

    def normalize(self, group="item"):
        if group == "item" and not self.opt._nrz_Q:
            self.Q = self._normalize(self.Q)
            self.opt._nrz_Q = True
        elif group == "user" and not self.opt._nrz_P:
            self.P = self._normalize(self.P)
            self.opt._nrz_P = True

    def initialize(self):
        super().initialize()
        self.init_factors()

    def init_factors(self):  # als.py:79-89
        assert self.data, "Data is not set"
        self.vdim = self.obj.get_vdim()
        header = self.data.get_header()
        for name, rows in [("P", header["num_users"]), ("Q", header["num_items"])]:
            setattr(self, name, np.abs(np.random.normal(scale=1.0 / (self.opt.d ** 2),
                                                        size=(rows, self.vdim)).astype("float32")))
        self.P[:, self.opt.d:] = 0.0
        self.Q[:, self.opt.d:] = 0.0
        self.obj.initialize_model(self.P, self.Q)

    def _get_buffer(self):
        buf = BufferedDataMatrix()
        buf.initialize(self.data)
        return buf

    def _iterate(self, buf, group="rowwise"):  # als.py:115-142
        int_group = 0 if group == "rowwise" else 1
        self.obj.precompute(int_group)
        loss_nume, loss_deno = 0.0, 0.0
        buf.set_group(group)
        for sz in buf.fetch_batch():
            start_x, next_x, indptr, keys, vals = buf.get()
            _n, _d = self.obj.partial_update(start_x, next_x, indptr, keys, vals, int_group)
            loss_nume += _n
            loss_deno += _d
        return loss_nume, loss_deno

    def train(self, training_callback=None):  # als.py:144-197
        self.obj.initialize_model(self.P, self.Q)
        buf = self._get_buffer()
        lindptr, rindptr, batch_size = buf.get_indptrs()
        self.obj.set_placeholder(lindptr, rindptr, batch_size)
        best_loss, rmse, self.validation_result = float("inf"), None, {}
        for i in range(self.opt.num_iters):
            start_t = time.time()
            n1, d1 = self._iterate(buf, group="rowwise")
            n2, d2 = self._iterate(buf, group="colwise")
            rmse = ((n1 + n2) / (d1 + d2 + self.opt.eps)) ** 0.5
            metrics = {"train_loss": rmse}
            if self.opt.validation and self.opt.evaluation_on_learning and \
               self.periodical(self.opt.evaluation_period, i):
                self.validation_result = self.get_validation_results()
                metrics.update({"val_%s" % k: v for k, v in self.validation_result.items()})
                if training_callback is not None and callable(training_callback):
                    training_callback(i, metrics)
            self.logger.info("Iteration %d: RMSE %.3f Elapsed %.3f secs" % (i + 1, rmse, time.time() - start_t))
            best_loss = self.save_best_only(rmse, best_loss, i)
            if self.early_stopping(rmse):
                break
        if self.opt.d < self.vdim:
            self.P = self.P[:, :self.opt.d]
            self.Q = self.Q[:, :self.opt.d]
        ret = {"train_loss": rmse}
        ret.update({"val_%s" % k: v for k, v in self.validation_result.items()})
        return ret

    def _get_data(self):
        data = super()._get_data()
        data.extend([("opt", self.opt), ("Q", self.Q), ("P", self.P)])
        return data

    def get_evaluation_metrics(self):
        return ["train_loss", "val_rmse", "val_ndcg", "val_map", "val_accuracy", "val_error"]
