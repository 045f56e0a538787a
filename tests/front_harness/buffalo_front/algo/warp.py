"""WARP front (mirror of /root/reference/buffalo/algo/warp.py; the reference refuses
accelerator=True at warp.py:31-32 -- this package provides that backend)."""
import numpy as np

from buffalo_amd.backend import CyWARP
from ._sgd_front import SgdFront
from .options import WARPOption


class WARP(SgdFront, WARPOption):
    NAME = "WARP"
    LOSS_NAME = "WARP-Loss"

    def __init__(self, opt_path=None, *args, **kwargs):
        WARPOption.__init__(self, *args, **kwargs)
        self._construct(opt_path, WARPOption, CyWARP, kwargs)
        # Q-23: the backend parsed score_func from the JSON before this lower-casing (warp.py:51-52)
        if isinstance(self.opt.score_func, str):
            self.opt.score_func = self.opt.score_func.lower()

    def init_factors(self):  # warp.py:79-92 (Q-18: signed N(0, 1/d^2); WARPOption has no use_bias)
        header = self.data.get_header()
        self.num_nnz = header["num_nnz"]
        d = self.opt.d
        self.P = np.random.normal(scale=1.0 / (d ** 2), size=(header["num_users"], d)).astype("float32")
        self.Q = np.random.normal(scale=1.0 / (d ** 2), size=(header["num_items"], d)).astype("float32")
        self.Qb = np.random.normal(scale=1.0 / (d ** 2), size=(header["num_items"], 1)).astype("float32")
        if not self.opt.use_bias:
            self.Qb *= 0
        self.obj.initialize_model(self.P, self.Q, self.Qb, self.num_nnz)
