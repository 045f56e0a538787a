"""The ALS accelerator surface (cuda/_als.pyx:25-67) evaluated in float64 numpy (TEST INFRASTRUCTURE): the SAME recurrence as
als.cc:107-209 + algo.cc:52-82 with every sum in double and the model kept in double between calls.  It is the yardstick of the
envelope tests: a backend's distance from this run is its own rounding, amplified by the conditioning of the case -- so a HIP
run is held to `err(hip, f64) <= 2.5 x err(oracle, f64)` where the oracle itself cannot follow the recurrence closely."""
import json

import numpy as np

import ref_numpy as rn


class F64ALS:
    def init(self, opt_path):
        path = opt_path.decode("utf-8") if isinstance(opt_path, bytes) else opt_path
        with open(path) as f:
            self.opt = json.load(f)
        assert self.opt["d"] < 128 and self.opt["optimizer"] in ("manual_cg", "llt", "ldlt"), "float64 stand-in: dense solves only"
        return True

    def get_vdim(self):
        return self.opt["d"]

    def set_placeholder(self, *args):
        pass

    def set_mode(self, *args):
        pass

    def initialize_model(self, P, Q):
        self.host = (P, Q)
        self.F = [P.astype(np.float64), Q.astype(np.float64)]

    def precompute(self, axis):
        Y = self.F[1 - axis]
        self.FF = Y.T @ Y

    def partial_update(self, start_x, next_x, indptr, keys, vals, axis):
        o = self.opt
        X, Y = self.F[axis], self.F[1 - axis]
        d = Y.shape[1]
        reg = o["reg_u"] if axis == 0 else o["reg_i"]
        shift = 0 if start_x == 0 else int(indptr[start_x - 1])
        nume = deno = 0.0
        closs = o["compute_loss_on_training"]
        for u in range(start_x, next_x):
            b = (0 if u == 0 else int(indptr[u - 1])) - shift
            e = int(indptr[u]) - shift
            if e == b:
                continue                                        # Q-16: empty rows stay as they are
            k, v = keys[b:e], vals[b:e].astype(np.float64)
            Ys = Y[k]
            ada = float(e - b) if o["adaptive_reg"] else 1.0
            p = X[u].copy()                                     # the loss terms see the row BEFORE its solve (als.cc:175-200)
            if closs:
                if axis == 1:
                    dot = Ys @ p
                    nume += float(p @ self.FF @ p) - float(dot @ dot) + float(((dot - 1.0) ** 2 * (1.0 + o["alpha"] * v)).sum())
                    deno += Y.shape[0] + float(o["alpha"] * v.sum())
                nume += ada * reg * float(p @ p)
            A = self.FF + o["alpha"] * (Ys * v[:, None]).T @ Ys + reg * ada * np.eye(d)
            y = ((1.0 + o["alpha"] * v)[:, None] * Ys).sum(axis=0)
            if o["optimizer"] == "manual_cg":
                X[u] = rn.manual_cg_f64(p, A, y, iters=o["num_cg_max_iters"], tol=o["cg_tolerance"], eps=o["eps"])
            else:
                X[u] = np.linalg.solve(A, y)
        self.host[axis][start_x:next_x] = X[start_x:next_x].astype(np.float32)
        return nume, deno
