"""BPRMF front (mirror of /root/reference/buffalo/algo/bpr.py)."""
import numpy as np

from buffalo_amd.backend import CyBPR
from ._sgd_front import SgdFront
from .options import BPRMFOption


class BPRMF(SgdFront, BPRMFOption):
    NAME = "BPRMF"
    LOSS_NAME = "PR-Loss"

    def __init__(self, opt_path=None, *args, **kwargs):
        BPRMFOption.__init__(self, *args, **kwargs)
        self._construct(opt_path, BPRMFOption, CyBPR, kwargs)

    def init_factors(self):  # bpr.py:84-97 (Q-18: |N(0, 1/d^2)|)
        header = self.data.get_header()
        self.num_nnz = header["num_nnz"]
        d = self.opt.d
        self.P = np.abs(np.random.normal(scale=1.0 / (d ** 2), size=(header["num_users"], d)).astype("float32"), order="C")
        self.Q = np.abs(np.random.normal(scale=1.0 / (d ** 2), size=(header["num_items"], d)).astype("float32"), order="C")
        self.Qb = np.abs(np.random.normal(scale=1.0 / (d ** 2), size=(header["num_items"], 1)).astype("float32"), order="C")
        if not self.opt.use_bias:
            self.Qb *= 0
        self.obj.initialize_model(self.P, self.Q, self.Qb, self.num_nnz)

    def prepare_sampling(self):  # bpr.py:99-111 incl. Q-4 (`int(sampling_power)`)
        header = self.data.get_header()
        self.sampling_table_ = np.zeros(header["num_items"], dtype=np.int64)
        if self.opt.sampling_power > 0.0:
            self.sampling_table_ += np.bincount(self.data.get_group("rowwise")["key"], minlength=header["num_items"])
            self.sampling_table_ **= int(self.opt.sampling_power)
            self.sampling_table_ = np.cumsum(self.sampling_table_)
        self.obj.set_cumulative_table(self.sampling_table_, header["num_items"])
