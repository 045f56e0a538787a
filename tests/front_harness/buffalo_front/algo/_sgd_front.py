"""Shared body of BPRMF and WARP (their train loops are identical in the reference:
/root/reference/buffalo/algo/bpr.py:170-250, warp.py:187-270)."""
import json
import time

import numpy as np

from .. import data as bdata
from ..data import BufferedDataMatrix, Data
from .base import Algo, Evaluable, get_logger


class SgdFront(Algo, Evaluable):
    NAME = "SGD"
    LOSS_NAME = "Loss"

    def _construct(self, opt_path, option_cls, backend_cls, kwargs):
        Algo.__init__(self)
        Evaluable.__init__(self)
        if opt_path is None:
            opt_path = option_cls().get_default_option()
        self.logger = get_logger(self.NAME)
        self.opt, self.opt_path = self.get_option(opt_path)
        if not self.opt.accelerator:
            raise NotImplementedError("buffalo_amd provides the accelerator (MI355X) backend only; "
                                      "set accelerator=True or use kakao/buffalo for the CPU path")
        self.obj = backend_cls()
        assert self.obj.init(bytes(self.opt_path, "utf-8")), "cannot parse option file: %s" % opt_path
        self.data = None
        data = kwargs.get("data")
        data_opt = kwargs.get("data_opt", self.opt.get("data_opt"))
        if data_opt:
            self.data = bdata.load(data_opt)
            self.data.create()
        elif isinstance(data, Data):
            self.data = data
        self.logger.info("%s(%s)" % (self.NAME, json.dumps(self.opt, indent=2)))
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
        assert self.data, "Data is not set"
        self.buf = BufferedDataMatrix()
        self.buf.initialize(self.data)
        self.init_factors()
        self.prepare_sampling()

    def prepare_sampling(self):
        pass

    def sampling_loss_samples(self):  # bpr.py:135-161 / warp.py:150-176
        users, positives, negatives = [], [], []
        if self.opt.compute_loss_on_training:
            header = self.data.get_header()
            num_loss_samples = int(header["num_users"] ** 0.5)
            _users = np.random.choice(range(self.P.shape[0]), size=num_loss_samples, replace=False)
            for u in _users:
                keys, *_ = self.data.get(u)
                if len(keys) == 0:
                    continue
                seen = set(keys)
                negs = np.random.choice(range(self.Q.shape[0]), size=len(seen) + 1, replace=False)
                negs = [n for n in negs if n not in seen]
                users.append(u)
                positives.append(keys[0])
                negatives.append(negs[0])
        self._sub_samples = [np.array(x, dtype=np.int32) for x in (users, positives, negatives)]

    def _iterate(self):  # bpr.py:170-188
        self.buf.set_group("rowwise")
        for sz in self.buf.fetch_batch():
            start_x, next_x, indptr, keys, _ = self.buf.get()
            self.obj.add_jobs(start_x, next_x, indptr, keys)
        self.obj.update_parameters()

    def compute_loss(self):
        if self._sub_samples[0].shape[0] == 0:
            return 0.0
        return self.obj.compute_loss(*self._sub_samples)

    def _prepare_train(self):  # bpr.py:195-209 (accelerator branch)
        vdim = self.obj.get_vdim()
        for attr in ["P", "Q"]:
            F = getattr(self, attr)
            if F.shape[1] < vdim:
                _F = np.empty(shape=(F.shape[0], vdim), dtype=np.float32)
                _F[:, :F.shape[1]] = F
                _F[:, self.opt.d:] = 0.0
                setattr(self, attr, _F)
        indptr, _, batch_size = self.buf.get_indptrs()
        self.obj.set_placeholder(indptr, batch_size)
        self.obj.initialize_model(self.P, self.Q, self.Qb, self.num_nnz, True)

    def _finalize_train(self):  # bpr.py:211-217 (the model came back with the last update_parameters, cuda/_bpr.pyx:59-60)
        self.P = self.P[:, :self.opt.d]
        self.Q = self.Q[:, :self.opt.d]
        return 0.0

    def train(self, training_callback=None):  # bpr.py:219-250
        self.validation_result = {}
        self.sampling_loss_samples()
        best_loss = float("inf")
        self._prepare_train()
        for i in range(self.opt.num_iters):
            start_t = time.time()
            self._iterate()
            self.obj.wait_until_done()
            loss = self.compute_loss() if self.opt.compute_loss_on_training else 0.0
            metrics = {"train_loss": loss}
            if self.opt.validation and self.opt.evaluation_on_learning and \
               self.periodical(self.opt.evaluation_period, i):
                self.validation_result = self.get_validation_results()
                metrics.update({"val_%s" % k: v for k, v in self.validation_result.items()})
                if training_callback is not None and callable(training_callback):
                    training_callback(i, metrics)
            self.logger.info("Iteration %s: %s %.3f Elapsed %.3f secs" % (i + 1, self.LOSS_NAME, loss, time.time() - start_t))
            best_loss = self.save_best_only(loss, best_loss, i)
            if self.early_stopping(loss):
                break
        ret = {"train_loss": self._finalize_train()}
        ret.update({"val_%s" % k: v for k, v in self.validation_result.items()})
        return ret

    def _get_data(self):
        data = super()._get_data()
        data.extend([("opt", self.opt), ("Q", self.Q), ("Qb", self.Qb), ("P", self.P)])
        return data

    def get_evaluation_metrics(self):
        return ["val_rmse", "val_ndcg", "val_map", "val_accuracy", "val_error", "train_loss"]
