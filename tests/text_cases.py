"""Lines for the text-parse tests: "row col val" in every decimal shape a rating file holds -- and the ones the device parser must hand back
to sscanf (near-ties of the double -> float rounding, > 19 digits, huge / tiny exponents, inf / nan / hex floats, ids beyond int)."""
import numpy as np


def value_tokens(rng, n):
    toks = ["1", "5", "0", "3.5", "4.0", "0.5", "10", "+2", "-1.5", "1e-3", "2.5E2", "0.333333", "1.0000001", "123456789", "0.000001",
            "16777216", "16777217", "33554433", "7.e1", ".25", "00012.50", "1e10", "1e-10", "9.999999e9", "12345678.9", "0.1", "0.2", "0.3",
            "4.35", "2.675", "1e22", "8.5e-23", "1.17549435e-38", "3.4028234e38",
            # handed back to the host: near-ties, long digit strings, out-of-range exponents, specials
            "1.00000005960464477539", "16777217.0000000001", "1.000000059604644775390625", "0.1000000014901161193847656250", "1e39", "1e-46",
            "4e-40", "inf", "-inf", "nan", "0x1p3", "123456789012345678901234567890", "1e", "2e+", "7.5e+1x"]
    out = list(toks)
    while len(out) < n:
        kind = rng.integers(0, 6)
        if kind == 0:
            out.append(str(int(rng.integers(0, 100000))))
        elif kind == 1:
            out.append(repr(float(np.float32(rng.random() * 10.0 ** rng.integers(-3, 4)))))
        elif kind == 2:
            out.append("%.6f" % (rng.random() * 5))
        elif kind == 3:
            out.append("%.3e" % (rng.random() * 10.0 ** rng.integers(-20, 20)))
        elif kind == 4:   # a float midpoint, printed exactly: the rounding to float is a tie / near-tie for the double route
            f = np.float32(rng.random() + 0.5)
            mid = (np.float64(f) + np.float64(np.nextafter(f, np.float32(4.0)))) / 2
            out.append("%.25f" % mid)
        else:
            out.append("%.17g" % (rng.random() * 10.0 ** rng.integers(-5, 6)))
    return out[:n]


def make_text(rng, n_lines, num_rows, num_cols, sorted_rows=False, sep_variants=True):
    """(text bytes, n_lines): ids 1-based; with `sorted_rows` the row ids are non-decreasing (the reference's keep-order mode needs that)."""
    rows = rng.integers(1, num_rows + 1, size=n_lines)
    if sorted_rows:
        rows = np.sort(rows)
    cols = rng.integers(1, num_cols + 1, size=n_lines)
    vals = value_tokens(rng, n_lines)
    rng.shuffle(vals)
    lines = []
    for k in range(n_lines):
        s1, s2 = " ", " "
        if sep_variants and k % 7 == 3:
            s1, s2 = "  ", "\t"
        tail = "\r" if sep_variants and k % 11 == 5 else (" " if sep_variants and k % 13 == 2 else "")
        lines.append("%d%s%d%s%s%s" % (rows[k], s1, cols[k], s2, vals[k], tail))
    text = "\n".join(lines)
    if n_lines % 2 == 0:
        text += "\n"          # with and without the final newline
    return text.encode(), n_lines
