"""Independent numpy / pure-python transliterations of the reference's training arithmetic.

These are the *pins* for the C++ oracle (SURVEY.md section 8c "Pins we must create ourselves"):
written separately from oracle/buffalo_oracle.cc, directly from the cited reference lines, in
float32 numpy.  Small cases only.  All paths relative to /root/reference/.
"""
import math

import numpy as np

f32 = np.float32


# ---------------------------------------------------------------------------------------------
# lib/algo_impl/bpr/bpr.cc:57-63, 124-131  (Q-2)
# ---------------------------------------------------------------------------------------------
def exp_table():
    t = np.zeros(1000, dtype=np.float32)
    for i in range(1000):
        x = f32(f32(f32(f32(i) / f32(1000)) * f32(2) - f32(1)) * f32(6))
        e = f32(np.exp(x))  # std::exp(float)
        t[i] = f32(1.0 / (float(e) + 1.0))
    return t


def bpr_logit(x, table):
    x = f32(x)
    if 6 < x:
        return f32(0.0)
    if x < -6:
        return f32(1.0)
    return table[int(f32(x + f32(6)) * f32(1000 // 6 // 2))]


# ---------------------------------------------------------------------------------------------
# lib/algo_impl/bpr/bpr.cc:119-171 -- one (u, pos, neg) SGD step, in place (Q-1)
# ---------------------------------------------------------------------------------------------
def bpr_sgd_step(P, Q, Qb, u, pos, neg, alpha, opt, table):
    alpha_d = alpha
    alpha = f32(alpha)
    reg_u, reg_i, reg_j, reg_b = (f32(opt[k]) for k in ("reg_u", "reg_i", "reg_j", "reg_b"))
    x = f32(np.dot(P[u], Q[pos] - Q[neg]))
    if opt["use_bias"]:
        x = f32(x + f32(Qb[pos, 0] - Qb[neg, 0]))
    logit = bpr_logit(x, table)
    item_deriv = (logit * P[u]).astype(np.float32)  # concrete: OLD P_u
    if opt["update_i"]:
        Q[pos] += alpha * (item_deriv - reg_i * Q[pos])
        if opt["use_bias"]:   # bpr.cc:161 is a scalar C++ statement: double alpha, double reg_b -> a double expression rounded once on store
            Qb[pos, 0] = f32(float(Qb[pos, 0]) + float(alpha_d) * (float(logit) - float(opt["reg_b"]) * float(Qb[pos, 0])))
    if opt["update_j"]:
        Q[neg] += alpha * (-item_deriv - reg_j * Q[neg])
        if opt["use_bias"]:   # bpr.cc:167
            Qb[neg, 0] = f32(float(Qb[neg, 0]) + float(alpha_d) * (-float(logit) - float(opt["reg_b"]) * float(Qb[neg, 0])))
    # lazy expression g evaluated now, with the UPDATED item rows
    g = logit * (Q[pos] - Q[neg]) - reg_u * P[u]
    P[u] += alpha * g
    return logit


# bpr.cc:138-156, 175-181 -- gradient accumulation branch (adam / adagrad)
def bpr_accumulate_step(P, Q, Qb, gP, gQ, gQb, u, pos, neg, opt, table):
    x = f32(np.dot(P[u], Q[pos] - Q[neg]))
    if opt["use_bias"]:
        x = f32(x + f32(Qb[pos, 0] - Qb[neg, 0]))
    logit = bpr_logit(x, table)
    item_deriv = (logit * P[u]).astype(np.float32)
    gP[u] += logit * (Q[pos] - Q[neg])
    if opt["update_i"]:
        gQ[pos] += item_deriv
        if opt["use_bias"]:
            gQb[pos] += logit
    if opt["update_j"]:
        gQ[neg] -= item_deriv
        if opt["use_bias"]:
            gQb[neg] -= logit
    return logit


# ---------------------------------------------------------------------------------------------
# lib/algo.cc:365-465 -- epoch-end optimizer pass (Q-5, Q-6, Q-9)
# ---------------------------------------------------------------------------------------------
def update_parameters(X, grad, mom, vel, cnt, reg, opt, iters, bias=False):
    """One factor matrix; mutates X, grad, mom, vel like the reference (grad keeps the step, Q-6)."""
    lr = f32(opt["lr"])
    beta1 = opt["beta1"]
    beta2 = opt["beta1"]  # sic: algo.cc:396
    r2 = f32(2 * reg)
    for r in range(X.shape[0]):
        if opt["per_coordinate_normalize"] and cnt is not None and cnt[r]:
            grad[r] = grad[r] / f32(cnt[r])
        grad[r] = grad[r] - X[r] * r2
        if opt["optimizer"] == "adam":
            mom[r] = f32(beta1) * mom[r] + f32(1.0 - beta1) * grad[r]
            vel[r] = f32(beta2) * vel[r] + f32(1.0 - beta2) * (grad[r] * grad[r])
            m_hat = mom[r] / f32(1.0 - beta1 ** (iters + 1))
            v_hat = vel[r] / f32(1.0 - beta2 ** (iters + 1))
            grad[r] = m_hat / (np.sqrt(v_hat) + f32(1e-10))
        else:
            vel[r] = vel[r] + grad[r] * grad[r]
            grad[r] = grad[r] / (np.sqrt(vel[r]) + f32(1e-10))
        X[r] = X[r] + lr * grad[r]


# ---------------------------------------------------------------------------------------------
# lib/algo_impl/warp/warp.cc:128-170 -- one positive (Q-10), dot score; draws from `draw(attempt)`
# ---------------------------------------------------------------------------------------------
def warp_positive(P, Q, gP, gQ, u, pos, seen, draw, opt):
    """Returns (accepted, neg, trial, n_scored)."""
    D = P.shape[1]
    max_trial = opt["max_trials"]
    thr = opt["threshold"]
    ui = f32(np.dot(P[u], Q[pos]))
    uj = f32(0)
    neg, trial, attempt, scored = 0, 1, 0, 0
    while trial <= max_trial:
        neg = draw(attempt)
        attempt += 1
        if neg in seen:
            continue
        trial += 1
        uj = f32(np.dot(P[u], Q[neg]))
        scored += 1
        if float(f32(ui - uj)) < thr:
            break
        trial += 1
    if trial >= max_trial:
        return False, neg, trial, scored
    Phi = f32(math.log(max(1, int((Q.shape[0] - len(seen) - 1) // trial))))
    ud = Phi * (Q[pos] - Q[neg])
    idv = Phi * P[u]
    gP[u] += ud - f32(opt["reg_u"]) * P[u]
    gQ[pos] += idv - f32(opt["reg_i"]) * Q[pos]
    gQ[neg] += -idv - f32(opt["reg_j"]) * Q[neg]
    assert D == gP.shape[1]
    return True, neg, trial, scored


def unit_ball_project(X):  # warp.cc:194-200 (Q-12)
    for r in range(X.shape[0]):
        n = max(f32(1.0), f32(np.sqrt(f32(np.dot(X[r], X[r])))))
        X[r] = X[r] / n


# ---------------------------------------------------------------------------------------------
# lib/algo_impl/als/als.cc:107-209 + lib/algo.cc:39-82
# ---------------------------------------------------------------------------------------------
def als_normal_equations(P, Q, FF, u, keys, vals, alpha, reg, adaptive_reg):
    """A = FF + alpha * sum v q q^T + reg*ada*I ;  y = sum (1 + v*alpha) q  (als.cc:180-202)."""
    D = Q.shape[1]
    A = np.zeros((D, D), dtype=np.float64)
    y = np.zeros(D, dtype=np.float64)
    for c, v in zip(keys, vals):
        q = Q[c].astype(np.float64)
        A += float(v) * np.outer(q, q)
        y += q * (1.0 + float(v) * alpha)
    A = FF.astype(np.float64) + alpha * A
    ada = float(len(keys)) if adaptive_reg else 1.0
    A += np.eye(D) * (reg * ada)
    return A, y


def manual_cg(x, A, y, iters=3, tol=1e-10, eps=1e-10):
    """lib/algo.cc:58-82 in float32 (Q-17)."""
    A = A.astype(np.float32)
    y = y.astype(np.float32)
    x = x.astype(np.float32).copy()
    r = y - x @ A
    if f32(np.dot(y, y)) < f32(np.dot(r, r)):
        x[:] = 0
        r = y.copy()
    p = r.copy()
    rs_old = f32(np.dot(r, r))
    for _ in range(iters):
        Ap = p @ A
        a = f32(rs_old / f32(f32(np.dot(Ap, p)) + f32(eps)))
        x = x + a * p
        r = r - a * Ap
        rs_new = f32(np.dot(r, r))
        if rs_new < f32(tol):
            break
        beta = f32(rs_new / f32(rs_old + f32(eps)))
        p = r + beta * p
        rs_old = rs_new
    return x


def als_loss_terms(P, Q, FF, u, keys, vals, alpha, reg, adaptive_reg, axis):
    """als.cc:175-178, 187-192, 198-200 -> (nume, deno) contribution of one row."""
    nume = deno = 0.0
    p = P[u].astype(np.float64)
    if axis == 1:
        nume += float(p @ (p @ FF.astype(np.float64)))
        deno += Q.shape[0]
        for c, v in zip(keys, vals):
            dot = float(np.dot(p, Q[c].astype(np.float64)))
            nume -= dot * dot
            nume += (dot - 1) ** 2 * (1.0 + float(v) * alpha)
            deno += float(v) * alpha
    ada = float(len(keys)) if adaptive_reg else 1.0
    nume += ada * reg * float(p @ p)
    return nume, deno


def ialspp_row(P, Q, FF, u, keys, vals, alpha, reg, block_size, tol=1e-10):
    """lib/algo_impl/als/als.cc:253-352 for a single row, float32 (Q-14). Returns new row."""
    D = Q.shape[1]
    P = P.astype(np.float32).copy()
    FF = FF.astype(np.float32)
    alpha, reg = f32(alpha), f32(reg)
    Yui = np.array([np.dot(P[u], Q[c]) for c in keys], dtype=np.float32)
    bs0 = min(D, block_size)
    for bb in range(0, D, bs0):
        bs = bs0
        if bb + bs >= D:
            bs = D - bb
        p_row = P[u].copy()
        gram = FF[:, bb:bb + bs]
        A = (gram[bb:bb + bs, :] + np.eye(bs, dtype=np.float32) * reg).astype(np.float32)
        b = (p_row @ gram + reg * p_row[bb:bb + bs]).astype(np.float32)
        for k, (c, v) in enumerate(zip(keys, vals)):
            residual = f32(Yui[k] - f32(1.0))
            b = b + f32(residual * f32(v) * alpha) * Q[c, bb:bb + bs]
        x = np.zeros(bs, dtype=np.float32)
        r = b.copy()
        p = r.copy()
        rsold = float(f32(np.dot(r, r)))
        if rsold > tol:
            for _ in range(3):
                Ap = (A @ p).astype(np.float32)
                for c, v in zip(keys, vals):
                    qb = Q[c, bb:bb + bs]
                    Ap = Ap + f32(f32(v) * alpha * f32(np.dot(qb, p))) * qb
                step = f32(rsold / float(f32(np.dot(p, Ap))))
                x = x + step * p
                r = r - step * Ap
                rsnew = float(f32(np.dot(r, r)))
                if rsnew < tol:
                    break
                p = r + f32(rsnew / rsold) * p
                rsold = rsnew
        P[u, bb:bb + bs] -= x
        for k, c in enumerate(keys):
            Yui[k] = f32(Yui[k] - f32(np.dot(Q[c, bb:bb + bs], x)))
    return P[u]


# ---------------------------------------------------------------------------------------------
# float64 ground truth of the same recurrences: fp32 implementations (oracle, HIP) are judged by
# how far they are from THIS, which separates rounding/conditioning from logic errors.
# ---------------------------------------------------------------------------------------------
def ialspp_row_f64(P, Q, FF, u, keys, vals, alpha, reg, block_size, tol=1e-10):
    D = Q.shape[1]
    p = P[u].astype(np.float64).copy()
    Qd, FFd = Q.astype(np.float64), FF.astype(np.float64)
    Y = np.array([p @ Qd[c] for c in keys])
    bs0 = min(D, block_size)
    for bb in range(0, D, bs0):
        bs = bs0 if bb + bs0 < D else D - bb
        gram = FFd[:, bb:bb + bs]
        A = gram[bb:bb + bs, :] + np.eye(bs) * reg
        b = p @ gram + reg * p[bb:bb + bs]
        for k, (c, v) in enumerate(zip(keys, vals)):
            b = b + (Y[k] - 1.0) * v * alpha * Qd[c, bb:bb + bs]
        x, r = np.zeros(bs), b.copy()
        pv, rsold = r.copy(), float(b @ b)
        if rsold > tol:
            for _ in range(3):
                Ap = A @ pv
                for c, v in zip(keys, vals):
                    qb = Qd[c, bb:bb + bs]
                    Ap = Ap + v * alpha * (qb @ pv) * qb
                step = rsold / (pv @ Ap)
                x, r = x + step * pv, r - step * Ap
                rsnew = float(r @ r)
                if rsnew < tol:
                    break
                pv, rsold = r + (rsnew / rsold) * pv, rsnew
        p[bb:bb + bs] -= x
        for k, c in enumerate(keys):
            Y[k] -= Qd[c, bb:bb + bs] @ x
    return p


def ialspp_row_f64_fast(p0, Qs, FF, vals, alpha, reg, block_size, tol=1e-10):
    """`ialspp_row_f64` with the per-entry loops written as matrix products (same recurrence, float64): `Qs` holds the rows
    of the other side this row touches, in entry order.  For rows of 1e5 entries, where the loop version takes minutes."""
    D = Qs.shape[1]
    p = p0.astype(np.float64).copy()
    Qd, FFd = Qs.astype(np.float64), FF.astype(np.float64)
    w = np.asarray(vals, dtype=np.float64) * alpha
    Y = Qd @ p
    bs0 = min(D, block_size)
    for bb in range(0, D, bs0):
        bs = bs0 if bb + bs0 < D else D - bb
        Qb = Qd[:, bb:bb + bs]
        A = FFd[bb:bb + bs, bb:bb + bs] + np.eye(bs) * reg
        b = p @ FFd[:, bb:bb + bs] + reg * p[bb:bb + bs] + ((Y - 1.0) * w) @ Qb
        x, r = np.zeros(bs), b.copy()
        pv, rsold = r.copy(), float(b @ b)
        if rsold > tol:
            for _ in range(3):
                Ap = A @ pv + Qb.T @ (w * (Qb @ pv))
                step = rsold / (pv @ Ap)
                x, r = x + step * pv, r - step * Ap
                rsnew = float(r @ r)
                if rsnew < tol:
                    break
                pv, rsold = r + (rsnew / rsold) * pv, rsnew
        p[bb:bb + bs] -= x
        Y -= Qb @ x
    return p


def manual_cg_f64(x0, A, y, iters=3, tol=1e-10, eps=1e-10):
    x = x0.astype(np.float64).copy()
    r = y - x @ A
    if y @ y < r @ r:
        x[:] = 0
        r = y.copy()
    p, rs = r.copy(), float(r @ r)
    for _ in range(iters):
        Ap = p @ A
        a = rs / (Ap @ p + eps)
        x, r = x + a * p, r - a * Ap
        rn = float(r @ r)
        if rn < tol:
            break
        p, rs = r + rn / (rs + eps) * p, rn
    return x


def als_half_epoch_f64(P, Q, FF, mat, opt, axis):
    """float64 result of one half-epoch for every row of `mat` (rows of P are updated)."""
    d = Q.shape[1]
    reg = opt["reg_u"] if axis == 0 else opt["reg_i"]
    optimizer = "ialspp" if d >= 128 else opt["optimizer"]
    T = P.astype(np.float64).copy()
    for u in range(mat.num_users):
        keys, vals = mat.row(u)
        if len(keys) == 0:
            continue
        if optimizer == "ialspp":
            T[u] = ialspp_row_f64(P, Q, FF, u, keys, vals, opt["alpha"], reg, opt["block_size"])
            continue
        A, y = als_normal_equations(P, Q, FF, u, keys, vals, opt["alpha"], reg, opt["adaptive_reg"])
        if optimizer == "manual_cg":
            T[u] = manual_cg_f64(P[u], A, y, iters=opt["num_cg_max_iters"], tol=opt["cg_tolerance"], eps=opt["eps"])
        else:
            T[u] = np.linalg.solve(A, y)
    return T


# ---------------------------------------------------------------------------------------------
# Published Philox4x32-10 (Salmon, Moraes, Dror, Shaw; SC'11) -- third, python-int implementation
# ---------------------------------------------------------------------------------------------
def philox4x32_10(ctr, key):
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & 0xffffffff, p1 & 0xffffffff, \
                         ((p0 >> 32) ^ c3 ^ k1) & 0xffffffff, p0 & 0xffffffff
        k0, k1 = (k0 + W0) & 0xffffffff, (k1 + W1) & 0xffffffff
    return [c0, c1, c2, c3]


# Random123 known-answer vectors (kat_vectors: philox4x32 10)
PHILOX_KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def counter_draw(seed, stream, pos_idx, slot, epoch, attempt):
    """The sampler contract shared by oracle (counter mode) and the HIP kernels."""
    return philox4x32_10((pos_idx & 0xffffffff, (pos_idx >> 32) & 0xffffffff, attempt,
                          ((epoch << 8) | (slot & 0xff)) & 0xffffffff),
                         (seed & 0xffffffff, 0x5bf03635 ^ stream))
