"""Thin re-creation of the pieces of buffalo.algo.base / buffalo.evaluate.base the three training
classes need (/root/reference/buffalo/algo/base.py:12-318, buffalo/evaluate/base.py:9-148).

Only what `train()` touches is reproduced here.  Ranking (validation, `topk_recommendation`) runs on the
GPU through `buffalo_amd.parallel`; id maps and the pickle-framed model format
are later rows of the scope table."""
import json
import logging
import time

import numpy as np

from ..misc import Option, load_option

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
        opt = load_option(opt_path)
        self.is_valid_option(opt)
        return Option(opt), opt_path

    def _open(self, name, backend_cls, opt_path, kwargs, data_types, accelerator_only=False):
        """What every front's constructor does after its option class is set up: option dict / path -> validated Option + the
        JSON file the backend is initialised from, the backend object, and the data (an option to load, or a Data object)."""
        from .. import data as bdata
        self.logger = get_logger(name)
        self.opt, self.opt_path = self.get_option(self.get_default_option() if opt_path is None else opt_path)
        if accelerator_only and not self.opt.accelerator:
            raise NotImplementedError("buffalo_amd provides the accelerator (MI355X) backend only; set accelerator=True or use "
                                      "kakao/buffalo for the CPU path")
        self.obj = backend_cls()
        assert self.obj.init(self.opt_path.encode("utf-8")), "cannot parse option file: %s" % opt_path
        self.data = None
        data_opt = kwargs.get("data_opt", self.opt.get("data_opt"))
        if data_opt:
            self.data = bdata.load(data_opt)
            self.data.create()
        elif isinstance(kwargs.get("data"), bdata.Data):
            self.data = kwargs["data"]
        self.logger.info("%s(%s)" % (name, json.dumps(self.opt, indent=2)))
        assert self.data is None or self.data.data_type in data_types, "%s trains on %s data" % (name, " / ".join(data_types))

    def set_data(self, data):
        from ..data import Data
        assert isinstance(data, Data), "Wrong instance: {}".format(type(data))
        self.data = data

    def _normalize_once(self, group, slots):
        """normalize() of every front: `slots` maps a group name to (factor attribute, flag in opt); a group is scaled once."""
        if group in slots:
            attr, flag = slots[group]
            if not self.opt.get(flag):
                setattr(self, attr, self._normalize(getattr(self, attr)))
                self.opt[flag] = True

    def _epochs(self, one_epoch, training_callback=None, prefix="val_", report=None):
        """The loop around every front's epoch: loss of the epoch, validation every `evaluation_period` (+ the caller's callback),
        best-model saving, early stopping.  Returns the last epoch's loss."""
        best, loss, self.validation_result = float("inf"), None, {}
        for i in range(self.opt.num_iters):
            began = time.time()
            loss = one_epoch(i)
            metrics = {"train_loss": loss}
            if self.opt.validation and self.opt.evaluation_on_learning and self.periodical(self.opt.evaluation_period, i):
                self.validation_result = self.get_validation_results()
                metrics.update({prefix + k: v for k, v in self.validation_result.items()})
                if callable(training_callback):
                    training_callback(i, metrics)
            if report:
                self.logger.info("Iteration %d: %s %.3f Elapsed %.3f secs" % (i + 1, report, loss, time.time() - began))
            best = self.save_best_only(loss, best, i)
            if self.early_stopping(loss):
                break
        return loss

    def _result(self, loss, prefix="val_"):
        out = {"train_loss": loss}
        out.update({prefix + k: v for k, v in self.validation_result.items()})
        return out

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

    def get_topk(self, scores, k, sorted=True, num_threads=4):
        """evaluate/base.py:31-42 (Evaluable.get_topk): column indices of the k best scores per row, on the GPU."""
        from buffalo_amd.parallel import quickselect
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        many = scores.ndim == 2
        if not many:
            scores = scores.reshape(1, -1)
        k = min(k, scores.shape[1])
        assert k > 0, f"k({k}) or cols({scores.shape[1]}) should be greater than 0"
        result = np.empty((scores.shape[0], k), dtype=np.int32)
        quickselect(scores, result, sorted, num_threads)
        return result if many else result[0]

    def _ranker(self):
        """TopK engine that admits every score -- what numpy scores + quickselect give in algo/base.py:40-55."""
        eng = getattr(self, "_topk_engine", None)
        if eng is None:
            from buffalo_amd.parallel import TopK
            eng = self._topk_engine = TopK()
            eng.set_mode("flt_min_rule", 0)
        return eng

    def _get_topk_recommendation(self, rows, topk, pool=None):
        """algo/base.py:40-55: (row, [item indices best first]) for every row; scores = P Q^T (+ Qb), fused with the selection."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        d = self.opt.d
        P = np.ascontiguousarray(self.P[:, :d], dtype=np.float32)
        Q = np.ascontiguousarray(self.Q[:, :d], dtype=np.float32)
        Qb = getattr(self, "Qb", None)
        Qb = np.ascontiguousarray(Qb, dtype=np.float32).reshape(-1, 1) if Qb is not None and getattr(self.opt, "use_bias", False) \
            else np.array([[]], dtype=np.float32)
        pool = np.array([], dtype=np.int32) if pool is None else np.ascontiguousarray(pool, dtype=np.int32)
        k = min(int(topk), Q.shape[0])
        keys = np.empty((len(rows), k), dtype=np.int32)
        scores = np.empty((len(rows), k), dtype=np.float32)
        self._ranker().dot_topn(rows, P, Q, Qb, keys, scores, pool, k)
        return list(zip(rows.tolist(), keys))

    def topk_recommendation(self, rows, topk=10, pool=None):
        """Index-based top-k (no id maps): {row: [item indices]}."""
        return {r: t for r, t in self._get_topk_recommendation(list(rows), topk, pool)}

    # -- id maps (base.py:157-180) ----------------------------------------------------------------
    def build_itemid_map(self):
        ids = list(getattr(self.data, "itemids", None) or map(str, range(self.data.get_header()["num_items"])))
        self._idmanager.itemids = ids
        self._idmanager.itemid_map = {v: idx for idx, v in enumerate(ids)}
        self._idmanager.itemid_mapped = True

    def build_userid_map(self):
        ids = list(getattr(self.data, "userids", None) or map(str, range(self.data.get_header()["num_users"])))
        self._idmanager.userids = ids
        self._idmanager.userid_map = {v: idx for idx, v in enumerate(ids)}
        self._idmanager.userid_mapped = True

    def get_index(self, keys, group="item"):
        """base.py:226-250: index (or list of indices) of the given key(s) in the id map of `group`; None where a key is unknown."""
        many = isinstance(keys, list)
        m = self._idmanager
        if group == "item":
            if not m.itemid_mapped:
                self.build_itemid_map()
            table = m.itemid_map
        elif group == "user":
            if not m.userid_mapped:
                self.build_userid_map()
            table = m.userid_map
        else:
            table = None
        found = [] if table is None else [table.get(k) for k in (keys if many else [keys])]
        return found if many else found[0]

    def get_index_pool(self, pool, group="item"):
        """base.py:252-268: a list of keys becomes the array of the indices that exist (unknown keys are DROPPED here, which is
        why the Par* classes' returned key list can slip against the indices); an ndarray passes through."""
        if isinstance(pool, list):
            return np.array([i for i in self.get_index(pool, group) if i is not None])
        if isinstance(pool, np.ndarray):
            return pool
        raise ValueError("Unexpected type for pool: %s" % type(pool))

    # -- Serializable (base.py:271-318): files are byte-compatible with stock buffalo, see buffalo_amd/serialize.py
    def save(self, path=None, with_itemid_map=True, with_userid_map=True, data_fields=()):
        from buffalo_amd.serialize import dump_objects
        if path is None:
            path = self.opt.model_path
        if with_itemid_map and not self._idmanager.itemid_mapped and getattr(self, "data", None) is not None:
            self.build_itemid_map()
        if with_userid_map and not self._idmanager.userid_mapped and getattr(self, "data", None) is not None:
            self.build_userid_map()
        data = self._get_data()
        if data_fields:
            data = [(k, v) for k, v in data if k in data_fields]
        dump_objects(path, data)

    def load(self, path, data_fields=()):
        from buffalo_amd.serialize import load_objects
        for name, obj in load_objects(path, data_fields):
            setattr(self, name, obj)

    @classmethod
    def instantiate(cls, cls_opt, path, data_fields=()):   # base.py:313-318
        c = cls(cls_opt().get_default_option())
        c.load(path, data_fields)
        return c

    def _get_data(self):
        return [("_idmanager", self._idmanager)]


class Evaluable:
    """Ranking / score metrics on the held-out `vali` group: evaluate/base.py:44-148 with the ranking done by
    the GPU top-k (`_get_topk_recommendation`) instead of numpy scores + OpenMP quickselect."""

    def __init__(self, *args, **kwargs):
        pass

    def get_validation_results(self, topk=None):
        if not self.data.has_group("vali"):
            return {}
        results = {}
        results.update(self._evaluate_ranking_metrics(topk))
        results.update(self._evaluate_score_metrics())
        return results

    def _validation_sets(self):
        """mm.py / base.py `_prepare_validation_data`: ground truth and already-seen items per validation row."""
        g = self.data.get_group("vali")
        gt = {}
        for r, c in zip(g["row"], g["col"]):
            gt.setdefault(int(r), set()).add(int(c))
        tr = self.data.get_group("rowwise")
        seen = {}
        for u in gt:
            beg = 0 if u == 0 else int(tr["indptr"][u - 1])
            seen[u] = set(int(x) for x in tr["key"][beg:int(tr["indptr"][u])])
        return gt, seen, max((len(s) for s in seen.values()), default=0)

    def _evaluate_ranking_metrics(self, topk=None):  # evaluate/base.py:44-128
        validation = self.opt.validation or {}
        batch_size = validation.get("batch", 128)
        topk = int(topk or validation.get("topk", 10))
        gt, validation_seen, max_seen = self._validation_sets()
        rows = np.array(sorted(gt), dtype=np.int32)
        num_items = self.data.get_header()["num_items"]
        NDCG = AP = HIT = AUC = N = 0.0
        idcgs = np.cumsum(1.0 / np.log2(np.arange(2, topk + 2)))
        dcgs = 1.0 / np.log2(np.arange(2, topk + 2))
        # the ranking asks for topk + (seen items of the batch's heaviest user) candidates; the GPU selection sorts up to
        # TOPK_LIMIT per row, so users whose history would not fit are ranked from dense scores on the host instead
        TOPK_LIMIT = 16384
        for index in range(0, len(rows), batch_size):
            batch = rows[index:index + batch_size]
            light = [r for r in batch if topk + len(validation_seen.get(int(r), ())) <= TOPK_LIMIT]
            heavy = [r for r in batch if topk + len(validation_seen.get(int(r), ())) > TOPK_LIMIT]
            need = topk + max((len(validation_seen.get(int(r), ())) for r in light), default=0)
            recs = self._get_topk_recommendation(np.array(light, dtype=np.int32), topk=min(need, num_items)) if light else []
            for r in heavy:
                d = self.opt.d
                sc = self.Q[:, :d] @ self.P[int(r), :d]
                if getattr(self.opt, "use_bias", False) and getattr(self, "Qb", None) is not None:
                    sc = sc + np.asarray(self.Qb).reshape(-1)
                recs.append((int(r), np.argsort(-sc, kind="stable")))
            for row, cand in recs:
                seen = validation_seen.get(row, set())
                if len(seen) == 0:
                    continue
                _topk = [int(t) for t in cand if int(t) not in seen][:topk]     # filter_seen_items
                _gt = gt[row]
                HIT += len(set(_topk) & _gt) / len(_gt)
                idcg = idcgs[min(len(_gt), topk) - 1]
                dcg = hit = miss = ap = auc = 0.0
                num_pos_items = len(_gt)
                num_neg_items = num_items - num_pos_items
                for i, r in enumerate(_topk):
                    if r in _gt:
                        hit += 1
                        ap += hit / (i + 1.0)
                        dcg += dcgs[i]
                    else:
                        miss += 1
                        auc += hit
                auc += ((hit + num_pos_items) / 2.0) * (num_neg_items - miss)
                auc /= (num_pos_items * num_neg_items)
                NDCG += dcg / idcg
                AP += ap / min(len(_gt), topk)
                AUC += auc
                N += 1.0
        if N == 0:
            return {}
        return {"ndcg": NDCG / N, "map": AP / N, "accuracy": HIT / N, "auc": AUC / N}

    def _get_scores(self, row, col):
        """als.py:106-108 / eals.py:100-102 / cfr.py:119-121: plain dot products (the SGD fronts override this)."""
        d = self.opt.d
        return (self.P[row, :d] * self.Q[col, :d]).sum(axis=1)

    def _evaluate_score_metrics(self):
        """evaluate/base.py:130-148, with its running sums as they are: entry by entry, in the scores' float32."""
        g = self.data.get_group("vali")
        scores = self._get_scores(g["row"], g["col"])
        error = rmse = 0.0
        for p, v in zip(scores, g["val"]):
            err = p - v
            error += abs(err)
            rmse += err * err
        return {"rmse": (rmse / len(scores)) ** 0.5, "error": error / len(scores)}
