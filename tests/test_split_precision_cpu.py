"""The arithmetic of the split-f16 ALS pass (csrc/als_kernels.hpp: als_split_f16, als_split_scale_kernel, `fused`) restated in numpy
and held to what DESIGN.md 4.5 claims for it -- no GPU:
  * x = S sqrt(w) q, S the power of two with S max|q| in [2^6, 2^7]; h = f16(x), l = f16(x - h) (round to nearest, remainder exact);
  * per 16 entries: acc += l^T h, acc += h^T l, acc += h^T h, products exact, sums kept in fp32;
  * the result / S^2 against the float64 Gramian, next to the fp32 instruction's arithmetic (w q rounded once, pairs of entries summed
    into an fp32 accumulator)."""
import numpy as np
import pytest


def _scale(q):
    qmax = float(np.abs(q).max())
    return 2.0 ** np.floor(np.log2(128.0 / qmax))


def _split(x):
    h = x.astype(np.float16)
    r = (x - h.astype(np.float32)).astype(np.float32)      # exact: h keeps x's leading bits
    assert np.array_equal(r.astype(np.float64), x.astype(np.float64) - h.astype(np.float64))
    l = r.astype(np.float16)
    return h.astype(np.float64), l.astype(np.float64)


def gram_split(q, w):
    S = np.float32(_scale(q))
    x = ((np.sqrt(w.astype(np.float32)) * S)[:, None] * q).astype(np.float32)
    assert np.abs(x).max() < 65504
    h, l = _split(x)
    d = q.shape[1]
    acc = np.zeros((d, d), np.float32)
    for k in range(0, len(w), 16):
        s = slice(k, k + 16)
        for t in (l[s].T @ h[s], h[s].T @ l[s], h[s].T @ h[s]):
            acc = (acc.astype(np.float64) + t).astype(np.float32)
    return acc.astype(np.float64) / (float(S) ** 2)


def gram_fp32(q, w):
    wq = (q * w[:, None].astype(np.float32)).astype(np.float32)
    d = q.shape[1]
    acc = np.zeros((d, d), np.float32)
    for k in range(0, len(w), 2):
        acc = (acc.astype(np.float64) + wq[k:k + 2].astype(np.float64).T @ q[k:k + 2].astype(np.float64)).astype(np.float32)
    return acc.astype(np.float64)


@pytest.mark.parametrize("n", [5, 16, 144, 2000])
@pytest.mark.parametrize("scale_q", [0.01, 1.0, 30.0])
def test_split_gramian_is_as_close_to_float64_as_the_fp32_instruction(n, scale_q):
    rng = np.random.default_rng(n)
    errs = []
    for rep in range(4):
        q = (rng.standard_normal((n, 128)) * scale_q).astype(np.float32)
        w = (8.0 * rng.integers(1, 6, size=n)).astype(np.float32)
        ref = (q.astype(np.float64).T * w.astype(np.float64)) @ q.astype(np.float64)
        nrm = np.abs(ref).max()
        errs.append((np.abs(gram_split(q, w) - ref).max() / nrm, np.abs(gram_fp32(q, w) - ref).max() / nrm))
    e_split, e_fp32 = np.mean([e[0] for e in errs]), np.mean([e[1] for e in errs])
    print("n %d |q| %g: split-f16 %.2e  fp32 instruction %.2e" % (n, scale_q, e_split, e_fp32))
    # rows shorter than a group pay the sqrt formulation's two extra roundings (3x at n = 5); from 16 entries on the two agree
    assert e_split <= (4.0 if n < 16 else 2.0) * e_fp32 + 1e-9


def test_window_of_weights_and_factors():
    """Weights over eight decades up to the cut (2^15) and factor entries down to 1e-3 of the largest stay inside f16's range with a
    full remainder: the Gramian is still at fp32's distance from float64."""
    rng = np.random.default_rng(1)
    n = 400
    q = rng.standard_normal((n, 128)).astype(np.float32)
    q[:, ::7] *= 1e-3                                     # small columns
    w = (10.0 ** rng.uniform(-3, np.log10(32768.0), size=n)).astype(np.float32)
    ref = (q.astype(np.float64).T * w.astype(np.float64)) @ q.astype(np.float64)
    got, base = gram_split(q, w), gram_fp32(q, w)
    nrm = np.abs(ref).max()
    assert np.abs(got - ref).max() / nrm <= 2.0 * np.abs(base - ref).max() / nrm + 1e-9
    # the small columns' own block, relative to ITS largest entry
    sub = np.ix_(range(0, 128, 7), range(0, 128, 7))
    assert np.abs(got[sub] - ref[sub]).max() / np.abs(ref[sub]).max() <= 4.0 * np.abs(base[sub] - ref[sub]).max() / np.abs(ref[sub]).max() + 1e-9


def test_scale_is_a_function_of_the_maximum_alone():
    """Any subset of rows that contains the largest entry -- a chunk, a rank's shard -- gives the same S (als_split_scale_kernel takes the
    maximum over the WHOLE other factor matrix, which every rank holds)."""
    rng = np.random.default_rng(2)
    q = rng.standard_normal((1000, 128)).astype(np.float32)
    S = _scale(q)
    assert 64.0 <= S * np.abs(q).max() < 128.0 + 1e-9
    assert _scale(q[::-1]) == S and _scale(np.concatenate([q[500:], q[:500]])) == S
