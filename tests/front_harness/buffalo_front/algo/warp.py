"""WARP front over `CyWARP` (stock buffalo's warp.py refuses accelerator = True at :31-32; its accelerator scaffold, :212-234, is
what this front follows)."""
from buffalo_amd.backend import CyWARP
from ._sgd_front import SgdFront
from .options import WARPOption


class WARP(SgdFront, WARPOption):
    NAME = "WARP"
    LOSS_NAME = "WARP-Loss"
    SIGNED_INIT = True

    def __init__(self, opt_path=None, *args, **kwargs):
        WARPOption.__init__(self, *args, **kwargs)
        self._construct(opt_path, CyWARP, kwargs)
        if isinstance(self.opt.score_func, str):     # Q-23: lower-cased only after the backend has parsed the JSON (warp.py:51-52)
            self.opt.score_func = self.opt.score_func.lower()
