"""What BPRMF and WARP share: stock buffalo's `bpr.py` / `warp.py` drive `CuBPR` (and the accelerator scaffold of WARP) the same
way -- loss samples, chunks of the row-wise matrix into `add_jobs`, one `update_parameters` per epoch.  Checked call by call
against the traces of the reference's own classes (tests/test_front_trace_cpu.py, cases bpr_* / warp_scaffold)."""
import numpy as np

from ..data import BufferedDataMatrix
from .base import Algo, Evaluable


class SgdFront(Algo, Evaluable):
    NAME = "SGD"
    LOSS_NAME = "Loss"
    SIGNED_INIT = False          # BPRMF starts from |N(0, 1/d^2)|, WARP from N(0, 1/d^2)

    def _construct(self, opt_path, backend_cls, kwargs):
        Algo.__init__(self)
        Evaluable.__init__(self)
        self._open(self.NAME, backend_cls, opt_path, kwargs, ["matrix"], accelerator_only=True)

    def normalize(self, group="item"):
        self._normalize_once(group, {"item": ("Q", "_nrz_Q"), "user": ("P", "_nrz_P")})

    def initialize(self):
        super().initialize()
        assert self.data, "Data is not set"
        self.buf = BufferedDataMatrix()
        self.buf.initialize(self.data)
        self.init_factors()
        self.prepare_sampling()

    def init_factors(self):  # bpr.py:84-97 / warp.py:79-92 (Q-18)
        header, d = self.data.get_header(), self.opt.d
        self.num_nnz = header["num_nnz"]
        for attr, shape in (("P", (header["num_users"], d)), ("Q", (header["num_items"], d)), ("Qb", (header["num_items"], 1))):
            F = np.random.normal(scale=1.0 / (d ** 2), size=shape).astype("float32")
            setattr(self, attr, F if self.SIGNED_INIT else np.abs(F, order="C"))
        if not self.opt.use_bias:
            self.Qb *= 0
        self.obj.initialize_model(self.P, self.Q, self.Qb, self.num_nnz)

    def prepare_sampling(self):
        pass

    def sampling_loss_samples(self):
        """sqrt(num_users) users, each with its first positive and one unseen item (bpr.py:135-161): the triples `compute_loss`
        is evaluated on after every epoch.  The draws go through np.random in the reference's order."""
        picked = []
        if self.opt.compute_loss_on_training:
            count = int(self.data.get_header()["num_users"] ** 0.5)
            for u in np.random.choice(range(self.P.shape[0]), size=count, replace=False):
                keys = self.data.get(u)[0]
                if len(keys):
                    seen = set(keys)
                    draws = np.random.choice(range(self.Q.shape[0]), size=len(seen) + 1, replace=False)
                    picked.append((u, keys[0], next(n for n in draws if n not in seen)))
        self._sub_samples = [np.array(col, dtype=np.int32) for col in (zip(*picked) if picked else ([], [], []))]

    def _get_scores(self, row, col):
        """bpr.py:131-133 / warp.py:145-150 as they are written: `Qb[col][0]` is the bias of the FIRST validation entry's item, added
        to every score; WARP's distance branch tests score_func == "L2" after the constructor lower-cased it (Q-23), so it is dead."""
        d = self.opt.d
        if self.opt.get("score_func") == "L2":
            return 1.0 - ((self.P[row, :d] - self.Q[col, :d]) ** 2).sum(-1)
        return (self.P[row, :d] * self.Q[col, :d]).sum(axis=1) + self.Qb[col][0]

    def compute_loss(self):
        return self.obj.compute_loss(*self._sub_samples) if len(self._sub_samples[0]) else 0.0

    def _widen_to_backend(self):
        """bpr.py:195-209: factors in the backend's padded width (pad columns zero), buffers announced, model handed over."""
        vdim, d = self.obj.get_vdim(), self.opt.d
        for attr in ("P", "Q"):
            F = getattr(self, attr)
            if F.shape[1] < vdim:
                W = np.empty((F.shape[0], vdim), dtype=np.float32)
                W[:, :F.shape[1]] = F
                W[:, d:] = 0.0
                setattr(self, attr, W)
        indptr, _, batch_size = self.buf.get_indptrs()
        self.obj.set_placeholder(indptr, batch_size)
        self.obj.initialize_model(self.P, self.Q, self.Qb, self.num_nnz, True)

    def _epoch(self, _):
        self.buf.set_group("rowwise")
        for _size in self.buf.fetch_batch():
            start_x, next_x, indptr, keys, _vals = self.buf.get()
            self.obj.add_jobs(start_x, next_x, indptr, keys)
        self.obj.update_parameters()
        self.obj.wait_until_done()
        return self.compute_loss() if self.opt.compute_loss_on_training else 0.0

    def train(self, training_callback=None):
        self.sampling_loss_samples()
        self._widen_to_backend()
        self._epochs(self._epoch, training_callback, report=self.LOSS_NAME)
        # bpr.py:211-217: the model came back with the last update_parameters (cuda/_bpr.pyx:59-60); the reported loss is 0
        self.P, self.Q = self.P[:, :self.opt.d], self.Q[:, :self.opt.d]
        return self._result(0.0)

    def _get_data(self):
        return super()._get_data() + [("opt", self.opt), ("Q", self.Q), ("Qb", self.Qb), ("P", self.P)]

    def get_evaluation_metrics(self):
        return ["val_rmse", "val_ndcg", "val_map", "val_accuracy", "val_error", "train_loss"]
