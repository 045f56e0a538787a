"""SPPMI matrix of a stream (CoFactor's context input): `bfh_sppmi_*` against the oracle's restatement of
stream.py:257-267 + fileio.hpp:109-254 + stream.py:169-195, and the oracle against a text-level transliteration.

Integer work (pairs, counts, row layout) is bit-exact.  Values: both sides evaluate the reference's double expression and
carry the six significant digits of its text output; an entry whose exact SPPMI is zero (cnt * D == app * app * k) has a
sign decided by the last bit of four logarithms and is allowed to be present on one side only."""
from collections import Counter
from fractions import Fraction
import math

import numpy as np
import pytest


def _stream(num_users, num_items, max_len, seed, skew=1.2):
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, max_len + 1, size=num_users)
    lens[rng.integers(0, num_users)] = 0
    lens[rng.integers(0, num_users)] = 1
    indptr = np.cumsum(lens).astype(np.int64)
    p = 1.0 / np.arange(1, num_items + 1) ** skew
    items = rng.choice(num_items, size=int(indptr[-1]), p=p / p.sum()).astype(np.int32)
    return indptr, items


def _text_level(indptr, items, num_items, windows, k):
    """The reference's flow on text lines, in Python: write the pair lines, group by first id, count, format with '%g',
    parse, sort by (row, col).  The group of the largest id is never written (fileio.hpp:182-250 flushes a group when the next
    id begins and not at end of file -- confirmed on the compiled reference in tests/test_oracle_ref_fileio.py)."""
    lines = []
    beg = 0
    for end in indptr:
        seq = [int(x) + 1 for x in items[beg:end]]
        beg = int(end)
        for i in range(len(seq)):
            for j in range(i + 1, i + windows + 1):
                if j >= len(seq):
                    break
                lines.append("%d %d" % (seq[i], seq[j]))
                lines.append("%d %d" % (seq[j], seq[i]))
    total = len(lines)
    parsed = sorted((tuple(int(t) for t in ln.split()) for ln in lines), key=lambda t: t[0])
    app = Counter(a for a, _ in parsed)
    groups = {}
    for a, c in parsed:
        groups.setdefault(a, []).append(c)
    out, exact_zero = [], set()
    eof_group = max(groups) if groups else None
    for probe, chunk in groups.items():
        if probe == eof_group:
            continue
        for c, cnt in Counter(chunk).items():
            if probe < c:
                continue
            if Fraction(cnt * total, app[probe] * app[c] * k) == 1:
                exact_zero.add((probe - 1, c - 1))
                exact_zero.add((c - 1, probe - 1))
            sppmi = math.log(cnt) + math.log(total) - math.log(app[probe]) - math.log(app[c]) - math.log(k)
            if sppmi > 0:
                txt = "%g" % sppmi
                out.append((probe - 1, c - 1, np.float32(txt)))
                out.append((c - 1, probe - 1, np.float32(txt)))
    out.sort(key=lambda t: (t[0], t[1]))
    return out, total, exact_zero


def _triples(g):
    rows = np.repeat(np.arange(len(g["indptr"])), np.diff(np.concatenate([[0], g["indptr"]])))
    return list(zip(rows.tolist(), g["key"].tolist(), g["val"].tolist()))


def _same_up_to_exact_zeros(got, want, exact_zero, ulp=0):
    gd, wd = Counter((r, c) for r, c, _ in got), Counter((r, c) for r, c, _ in want)
    for rc in set(gd) | set(wd):
        assert gd[rc] == wd[rc] or rc in exact_zero, rc
    gv, wv = {(r, c): v for r, c, v in got}, {(r, c): v for r, c, v in want}
    for rc in set(gv) & set(wv):
        a, b = np.float32(gv[rc]), np.float32(wv[rc])
        assert a == b or (ulp and abs(int(a.view(np.int32)) - int(b.view(np.int32))) <= ulp), (rc, a, b)


@pytest.mark.parametrize("num_users,num_items,max_len,windows,k,seed", [(60, 25, 14, 3, 1, 0), (200, 90, 30, 5, 2, 1), (40, 7, 9, 2, 1, 2),
                                                                       (5, 300, 40, 50, 1, 3)])
def test_oracle_follows_the_text_level_flow(oracle, num_users, num_items, max_len, windows, k, seed):
    indptr, items = _stream(num_users, num_items, max_len, seed)
    g = oracle.build_sppmi(indptr, items, num_items, windows, k)
    want, total, exact_zero = _text_level(indptr, items, num_items, windows, k)
    assert g["total_lines"] == total
    got = _triples(g)
    assert [t[:2] for t in got] == sorted(t[:2] for t in got)              # rows ascending, columns ascending inside a row
    _same_up_to_exact_zeros(got, want, exact_zero, ulp=1)                   # numpy parses text through a double
    if any(r == c for r, c, _ in got):
        assert Counter((r, c) for r, c, _ in got)[next((r, c) for r, c, _ in got if r == c)] == 2   # "p p v" is written twice


def _device_formulation(indptr, items, num_items, windows, k):
    """What csrc/sppmi.hip does, step for step in numpy: every pair line as the key first * I + second, sorted; distinct
    keys with their counts; appearances of an id = keys that start with it; one value per distinct key from the reference's double
    expression with probe = the larger id; keys whose probe is the FIRST ID OF THE LAST SORTED KEY -- the group stock buffalo never
    writes -- dropped; entry (a, b) from its own key (twice when a == b); six-digit text round trip."""
    keys = []
    beg = 0
    for end in indptr:
        seq = [int(x) for x in items[beg:end]]
        beg = int(end)
        for i in range(len(seq)):
            for j in range(i + 1, min(i + windows + 1, len(seq))):
                keys += [seq[i] * num_items + seq[j], seq[j] * num_items + seq[i]]
    if not keys:
        return []
    keys = np.sort(np.asarray(keys, dtype=np.int64))
    uniq, cnt = np.unique(keys, return_counts=True)
    app = np.bincount(keys // num_items, minlength=num_items)
    eof_group = int(uniq[-1] // num_items)
    log_d, log_k = math.log(len(keys)), math.log(k)
    out = []
    for key, c in zip(uniq.tolist(), cnt.tolist()):
        a, b = divmod(key, num_items)
        probe, other = max(a, b), min(a, b)
        sppmi = math.log(c) + log_d - math.log(app[probe]) - math.log(app[other]) - log_k
        if sppmi > 0 and probe != eof_group:
            out += [(a, b, np.float32("%g" % sppmi))] * (2 if a == b else 1)
    return out


@pytest.mark.parametrize("num_users,num_items,max_len,windows,k,seed", [(60, 25, 14, 3, 1, 0), (200, 90, 30, 5, 2, 1), (40, 7, 9, 2, 1, 2),
                                                                       (5, 300, 40, 50, 1, 3), (1, 10, 12, 3, 1, 5)])
def test_the_device_s_formulation_is_the_oracle_s(oracle, num_users, num_items, max_len, windows, k, seed):
    """No GPU here: the formulation the kernels implement, restated in numpy, against the oracle -- in particular the rule that finds
    the reference's unwritten end-of-file group from the sorted keys alone."""
    indptr, items = _stream(num_users, num_items, max_len, seed)
    got = _device_formulation(indptr, items, num_items, windows, k)
    want = _triples(oracle.build_sppmi(indptr, items, num_items, windows, k))
    _, _, exact_zero = _text_level(indptr, items, num_items, windows, k)
    assert [t[:2] for t in got] == sorted(t[:2] for t in got)       # already in (row, col) order: no second sort on the device
    _same_up_to_exact_zeros(got, want, exact_zero, ulp=1)
    if not exact_zero:
        assert [t[:2] for t in got] == [t[:2] for t in want]


def test_oracle_edge_cases(oracle):
    g = oracle.build_sppmi(np.array([0, 1, 1], np.int64), np.array([3], np.int32), 5, 4, 1)      # nobody has two events
    assert g["total_lines"] == 0 and len(g["key"]) == 0 and np.all(g["indptr"] == 0)
    g = oracle.build_sppmi(np.array([2], np.int64), np.array([1, 1], np.int32), 3, 1, 1)         # one self pair: lines "2 2" twice
    assert g["total_lines"] == 2 and len(g["key"]) == 0                                          # log(2)+log(2)-log(2)-log(2) = 0: not > 0


@pytest.mark.gpu
@pytest.mark.parametrize("num_users,num_items,max_len,windows,k,seed", [(60, 25, 14, 3, 1, 0), (200, 90, 30, 5, 2, 1), (40, 7, 9, 2, 1, 2),
                                                                       (5, 300, 40, 50, 1, 3), (3000, 1500, 60, 5, 1, 4), (1, 10, 12, 3, 1, 5),
                                                                       (800, 70000, 25, 4, 1, 6)])     # > 65,536 items: 64-bit keys
def test_device_matches_oracle(oracle, num_users, num_items, max_len, windows, k, seed):
    from buffalo_amd import ingest
    indptr, items = _stream(num_users, num_items, max_len, seed)
    g = ingest.build_sppmi(indptr, items, num_items, windows, k)
    o = oracle.build_sppmi(indptr, items, num_items, windows, k)
    assert g["total_lines"] == o["total_lines"]
    if np.array_equal(g["indptr"], o["indptr"]) and np.array_equal(g["key"], o["key"]):
        assert np.array_equal(g["val"].view(np.int32), o["val"].view(np.int32))                 # bit-identical values
        return
    _, _, exact_zero = _text_level(indptr, items, num_items, windows, k)
    _same_up_to_exact_zeros(_triples(g), _triples(o), exact_zero)
    got = _triples(g)
    assert [t[:2] for t in got] == sorted(t[:2] for t in got)


@pytest.mark.gpu
def test_device_edge_cases_and_errors():
    from buffalo_amd import ingest
    from buffalo_amd._lib import BuffaloHipError
    g = ingest.build_sppmi(np.array([0, 1, 1], np.int64), np.array([3], np.int32), 5, 4, 1)
    assert g["total_lines"] == 0 and len(g["key"]) == 0 and np.all(g["indptr"] == 0)
    with pytest.raises(BuffaloHipError):
        ingest.build_sppmi(np.array([2], np.int64), np.array([1, 9], np.int32), 3, 1, 1)         # item id outside the catalogue
    with pytest.raises(BuffaloHipError):
        ingest.build_sppmi(np.array([2], np.int64), np.array([1, 2], np.int32), 3, 0, 1)         # windows must be positive


@pytest.mark.gpu
def test_stream_sized_properties():
    """ML-20M-sized stream (138,493 sequences, 20 M events, windows 5: 200 M lines): symmetric, sorted, values inside
    (0, log D], a sampled row checked against a direct count."""
    import bench
    from buffalo_amd import ingest
    csr = bench.load_matrix("ml20m", 7)
    rng = np.random.default_rng(0)
    items = csr.keys.copy()
    # a sequence order: shuffle every user's items (the matrix stores them sorted)
    order = np.argsort(np.repeat(np.arange(csr.num_users), np.diff(np.concatenate([[0], csr.indptr]))) + rng.random(csr.nnz))
    items = items[order]
    g, st = ingest.build_sppmi(csr.indptr, items, csr.num_items, 5, 1, with_stats=True)
    print("sppmi of %d events: %d lines, %d distinct pairs, nnz %d, device %.1f ms" % (csr.nnz, g["total_lines"], st["launches"], len(g["key"]), st["kernel_ms"]))
    lens = np.diff(np.concatenate([[0], csr.indptr]))
    assert g["total_lines"] == int(2 * sum(np.where(lens <= 6, lens * (lens - 1) // 2, (lens - 5) * 5 + 10)))
    assert np.all(np.diff(g["indptr"]) >= 0) and g["indptr"][-1] == len(g["key"])
    assert np.all(g["val"] > 0) and np.all(g["val"] <= np.log(g["total_lines"]) + 1e-3)
    rows = np.repeat(np.arange(csr.num_items), np.diff(np.concatenate([[0], g["indptr"]])))
    key64 = rows.astype(np.int64) * csr.num_items + g["key"]
    assert np.all(np.diff(key64) >= 0)
    sym = np.sort(g["key"].astype(np.int64) * csr.num_items + rows)
    assert np.array_equal(sym, key64)                                       # (a, b) listed <=> (b, a) listed
