"""Thin re-creation of the pieces of buffalo.algo.base / buffalo.evaluate.base the three training
classes need (/root/reference/buffalo/algo/base.py:12-318, buffalo/evaluate/base.py:9-148).

Inference helpers (top-k, most_similar), id maps and the pickle-framed model format are NEXT rows of
the scope table; only what `train()` touches is reproduced here, plus a plain
numpy `topk_recommendation` so examples and parity checks can rank."""
import logging
import pickle
import struct

import numpy as np

from ..misc import Option

EPS = 1e-8


def get_logger(name):
    return logging.getLogger("buffalo_amd." + name)


class Algo:
    def __init__(self, *args, **kwargs):
        self._idmanager = Option({"userid": [], "userid_map": {}, "itemid": [], "itemid_map": {},
                                  "userid_mapped": False, "itemid_mapped": False})

    def get_option(self, opt_path):  # base.py:19-26
        if isinstance(opt_path, (dict, Option)):
            opt_path = self.create_temporary_option_from_dict(opt_path)
        opt = Option(opt_path)
        self.is_valid_option(opt)
        return Option(opt), opt_path

    def _normalize(self, feat):  # base.py:28-30
        return feat / np.sqrt((feat ** 2).sum(-1) + EPS)[..., np.newaxis]

    def initialize(self):  # base.py:32-36
        self.__early_stopping = {"round": 0, "min_loss": 987654321}
        if self.opt.random_seed:
            np.random.seed(self.opt.random_seed)

    def periodical(self, period, current):  # base.py:202-205
        return not period or (current + 1) % period == 0

    def save_best_only(self, loss, best_loss, i):  # base.py:207-211
        if self.opt.save_best and best_loss > loss and self.periodical(self.opt.save_period, i):
            self.save(self.opt.model_path)
            return loss
        return best_loss

    def early_stopping(self, loss):  # base.py:213-224
        if self.opt.early_stopping_rounds < 1:
            return False
        if self.__early_stopping["min_loss"] < loss:
            self.__early_stopping["round"] += 1
        else:
            self.__early_stopping["round"] = 0
        self.__early_stopping["min_loss"] = loss
        return self.__early_stopping["round"] >= self.opt.early_stopping_rounds

    def get_topk(self, scores, k):
        k = min(k, scores.shape[-1])
        part = np.argpartition(-scores, k - 1, axis=-1)[..., :k]
        order = np.argsort(-np.take_along_axis(scores, part, axis=-1), axis=-1)
        return np.take_along_axis(part, order, axis=-1)

    def topk_recommendation(self, rows, topk=10):
        """Index-based top-k (no id maps): {row: [item indices]}."""
        rows = list(rows)
        scores = self.P[rows] @ self.Q.T
        Qb = getattr(self, "Qb", None)
        if Qb is not None and getattr(self.opt, "use_bias", False):
            scores = scores + Qb.reshape(1, -1)
        return dict(zip(rows, self.get_topk(scores, topk)))

    # -- Serializable (base.py:271-318): u64 count, then (u64 len, name, u64 len, pickle) frames ----
    def save(self, path):
        data = self._get_data()
        with open(path, "wb") as fout:
            fout.write(struct.pack("Q", len(data)))
            for name, obj in data:
                nb, ob = name.encode("utf8"), pickle.dumps(obj, protocol=4)
                fout.write(struct.pack("Q", len(nb)) + nb + struct.pack("Q", len(ob)) + ob)

    def load(self, path, data_fields=()):
        with open(path, "rb") as fin:
            (n,) = struct.unpack("Q", fin.read(8))
            for _ in range(n):
                (ln,) = struct.unpack("Q", fin.read(8))
                name = fin.read(ln).decode("utf8")
                (lo,) = struct.unpack("Q", fin.read(8))
                blob = fin.read(lo)
                if not data_fields or name in data_fields:
                    setattr(self, name, pickle.loads(blob))

    def _get_data(self):
        return [("_idmanager", self._idmanager)]


class Evaluable:
    """Ranking metrics on the held-out `vali` group (evaluate/base.py:44-148, numpy restatement)."""

    def __init__(self, *args, **kwargs):
        pass

    def get_validation_results(self, topk=10):
        if not self.data.has_group("vali"):
            return {}
        g = self.data.get_group("vali")
        rows, cols = g["row"], g["col"]
        users = np.unique(rows)
        scores = self.P[users][:, :self.opt.d] @ self.Q[:, :self.opt.d].T
        Qb = getattr(self, "Qb", None)
        if Qb is not None and getattr(self.opt, "use_bias", False):
            scores = scores + Qb.reshape(1, -1)
        tr = self.data.get_group("rowwise")
        pos = {u: i for i, u in enumerate(users)}
        for u in users:  # seen items never get recommended
            beg = 0 if u == 0 else int(tr["indptr"][u - 1])
            scores[pos[u], tr["key"][beg:int(tr["indptr"][u])]] = -np.inf
        top = self.get_topk(scores, topk)
        truth = {}
        for r, c in zip(rows, cols):
            truth.setdefault(int(r), set()).add(int(c))
        ndcg = mapk = acc = 0.0
        idcgs = np.cumsum(1.0 / np.log2(np.arange(2, topk + 2)))
        for u in users:
            gt, rec = truth[int(u)], top[pos[u]]
            hits = np.array([int(x) in gt for x in rec], dtype=np.float64)
            dcg = (hits / np.log2(np.arange(2, len(rec) + 2))).sum()
            ndcg += dcg / idcgs[min(len(gt), topk) - 1]
            prec = np.cumsum(hits) / np.arange(1, len(rec) + 1)
            mapk += (prec * hits).sum() / min(len(gt), topk)
            acc += hits.sum() / min(len(gt), topk)
        n = float(len(users))
        pred = np.array([self.P[r, :self.opt.d] @ self.Q[c, :self.opt.d] for r, c in zip(rows, cols)])
        rmse = float(np.sqrt(np.mean((pred - g["val"]) ** 2)))
        return {"ndcg": ndcg / n, "map": mapk / n, "accuracy": acc / n, "rmse": rmse, "error": rmse}
