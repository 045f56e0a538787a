"""BPRMF front over `CyBPR` (stock buffalo: buffalo/algo/bpr.py with accelerator = True)."""
import numpy as np

from buffalo_amd.backend import CyBPR
from ._sgd_front import SgdFront
from .options import BPRMFOption


class BPRMF(SgdFront, BPRMFOption):
    NAME = "BPRMF"
    LOSS_NAME = "PR-Loss"

    def __init__(self, opt_path=None, *args, **kwargs):
        BPRMFOption.__init__(self, *args, **kwargs)
        self._construct(opt_path, CyBPR, kwargs)

    def prepare_sampling(self):
        """bpr.py:99-111: cumulative item-popularity table of the negative sampler (all zero = uniform); the exponent is
        truncated to an integer as the reference truncates it (Q-4)."""
        items = self.data.get_header()["num_items"]
        table = np.zeros(items, dtype=np.int64)
        if self.opt.sampling_power > 0.0:
            table += np.bincount(self.data.get_group("rowwise")["key"], minlength=items)
            table **= int(self.opt.sampling_power)
            table = np.cumsum(table)
        self.sampling_table_ = table
        self.obj.set_cumulative_table(table, items)
