"""Shared top-k fixtures: the reference's own parallel tests (tests/parallel/test_base.py) restated, and
exact-arithmetic cases (small-integer factors: every dot product is exact in fp32 in any summation
order) that pin the admission / tie rules of parallel::dot_topn bit for bit."""
import numpy as np

EMPTY_POOL = np.array([], dtype=np.int32)
NO_BIAS = np.array([[]], dtype=np.float32)


def unit_factors(rows, cols, seed):
    """tests/parallel/test_base.py:14-17 (get_factors) with a seeded generator."""
    rng = np.random.default_rng(seed)
    F = rng.random((rows, cols)).astype(np.float32)
    return (F / np.sqrt((F ** 2).sum(-1) + 1e-8)[..., np.newaxis]).astype(np.float32)


def numpy_most_similar(indexes, F, topk):
    """tests/parallel/test_base.py:19-25."""
    topk += 1
    scores = F[indexes].dot(F.T)
    topks = np.argsort(scores, axis=1)[:, -topk:][:, ::-1]
    topks = np.array([t[1:] for t in topks])
    return topks, np.array([s[t] for t, s in zip(topks, scores)])


def numpy_topk(indexes, P, F, topk):
    """tests/parallel/test_base.py:27-31."""
    scores = P[indexes].dot(F.T)
    topks = np.argsort(scores, axis=1)[:, -topk:][:, ::-1]
    return topks, np.array([s[t] for t, s in zip(topks, scores)])


def integer_factors(rows, cols, seed, lo=-2, hi=3):
    return np.random.default_rng(seed).integers(lo, hi, size=(rows, cols)).astype(np.float32)


def run(fn, indexes, P, Q, Qb, pool, k):
    out_keys = np.full((len(indexes), k), 12345, dtype=np.int32)
    out_scores = np.full((len(indexes), k), 9.75, dtype=np.float32)
    fn(np.ascontiguousarray(indexes, dtype=np.int32), P, Q, Qb, out_keys, out_scores, np.ascontiguousarray(pool, dtype=np.int32), k, 1)
    return out_keys, out_scores


def spec_dot_topn(indexes, P, Q, Qb, pool, k, same):
    """Closed form of the reference's running list (_core.hpp:37-67, 115-128), independent of the oracle's
    insertion loop (float64 scores are exact for the integer cases).  Only scores > FLT_MIN are ever admitted.
    With t the kk-th largest admissible score: every candidate > t is kept; a candidate == t is admitted only
    while fewer than kk candidates >= t have been seen (F = the first kk such candidates, A = the ties in F),
    and every better candidate arriving after F evicts the OLDEST tie -- so the surviving ties are the
    highest-index members of A.  Listed by (score desc, index desc); unfilled slots (-1, FLT_MIN)."""
    fmin = np.finfo(np.float32).tiny
    q_rows = Q.shape[0]
    kk = min(q_rows, k)
    if len(pool):
        kk = min(len(pool), kk)
    allowed = set(int(x) for x in pool)
    keys = np.full((len(indexes), k), -1, np.int32)
    vals = np.zeros((len(indexes), k), np.float32)
    for i, q in enumerate(indexes):
        cand = []
        for j in range(q_rows):
            if same and j == q:
                continue
            if allowed and j not in allowed:
                continue
            s = np.float32(np.dot(P[q].astype(np.float64), Q[j].astype(np.float64)) + (float(Qb[j, 0]) if Qb.shape[1] else 0.0))
            if s > fmin:
                cand.append((s, j))
        if len(cand) <= kk:
            kept = cand
        else:
            t = sorted((c[0] for c in cand), reverse=True)[kk - 1]
            better = [c for c in cand if c[0] > t]
            first = [c for c in cand if c[0] >= t][:kk]            # F (cand is in index order)
            ties = [c for c in first if c[0] == t]                  # A
            kept = better + ties[len(ties) - (kk - len(better)):]
        kept = sorted(kept, key=lambda c: (-c[0], -c[1]))
        for r in range(kk):
            if r < len(kept):
                keys[i, r], vals[i, r] = kept[r][1], kept[r][0]
            else:
                keys[i, r], vals[i, r] = -1, fmin
    return keys, vals
