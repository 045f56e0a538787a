"""A/B of the config-#3 ALS row kernel: round 3's wave-per-row split-f16 kernel (als_pc=0) against the producer / consumer pairs
(als_pc=1, als_pc.hpp): kernel time per half-epoch, ablations of the new kernel (als_debug bits), agreement of the two results from
one warm state, and the SIMD-placement statistic of the role assignment.
    python scripts/als_pc_ab.py [--ablate]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from buffalo_amd import ingest, synth
from buffalo_amd.backend import CyALS

csr = bench.load_matrix("ml20m", 7)
U, I, nnz = csr.num_users, csr.num_items, csr.nnz
vals = (1 + np.random.default_rng(7).poisson(1.0, size=nnz)).astype(np.float32)
col = ingest.coo_to_csr(csr.keys, csr.rows(), vals, I, U)
D = int(os.environ.get("ALS_D", bench.D))                    # 64 / 96: the pc kernel at T = 2 / 3 (iALS++ has to be asked for below d = 128)
OPT = dict(bench.ALS_OPT, d=D, optimizer="ialspp")


def make(P, Q, modes):
    g = CyALS()
    path = bench._opt_file(OPT)
    assert g.init(path)
    os.unlink(path)
    g.initialize_model(P, Q)
    g.set_resident_csr(0, csr.indptr, csr.keys, vals)
    g.set_resident_csr(1, col["indptr"], col["key"], col["val"])
    for k, v in modes.items():
        g.set_mode(k, v)
    return g


def half(g, axis):
    rows, ip = (U, csr.indptr) if axis == 0 else (I, col["indptr"])
    g.precompute(axis)
    g.reset_stats()
    g.partial_update(0, rows, ip, None, None, axis)
    return g.stats()["kernel_ms"]


def timing(modes, epochs=4):
    P, Q, _ = synth.init_factors(U, I, D, seed=7)
    g = make(P, Q, dict(modes, als_writeback=0))
    per = {0: [], 1: []}
    for ep in range(epochs):
        for axis in (0, 1):
            ms = half(g, axis)
            if ep:
                per[axis].append(ms)
    extra = ""
    if modes.get("als_pc", 1):
        try:
            extra = "  workgroups with one pair per SIMD: %d of %d" % (g.device_buffer("als_pc_same_simd")[1], 256)
        except Exception as e:   # noqa
            extra = "  (%s)" % e
    print("%-40s user half-epoch %.3f ms  item half-epoch %.3f ms  sum %.3f%s" % (modes, np.mean(per[0]), np.mean(per[1]), np.mean(per[0]) + np.mean(per[1]), extra), flush=True)


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


for m in ({"als_pc": 0}, {"als_pc": 1}):
    timing(m)
if "--ablate" in sys.argv:   # results are wrong with these, timings only
    for bits, what in ((1, "no block solve"), (16, "no matrix instructions"), (17, "neither"), (17 + 32, "neither, producer without arithmetic"),
                       (512, "matrix instructions free-running (not waiting for pieces)"), (512 + 1, "... and no solve")):
        print("als_debug %d (%s):" % (bits, what), end=" ")
        timing({"als_pc": 1, "als_debug": bits}, epochs=3)

if "--timing-only" in sys.argv:
    sys.exit(0)
# the same warm state through both kernels
P, Q, _ = synth.init_factors(U, I, D, seed=7)
g = make(P, Q, {"als_pc": 0})
for ep in range(2):
    half(g, 0), half(g, 1)
g.synchronize(True)
Pw, Qw = P.copy(), Q.copy()
del g
out = {}
for name, m in (("wave", {"als_pc": 0}), ("pc", {"als_pc": 1})):
    P1, Q1 = Pw.copy(), Qw.copy()
    g = make(P1, Q1, m)
    half(g, 0)
    g.synchronize(True)
    Pmid = P1.copy()
    half(g, 1)
    g.synchronize(True)
    out[name] = (Pmid, Q1.copy())
    del g
print("one epoch from the same warm state, producer/consumer vs wave-per-row: P max-rel %.3e   Q max-rel %.3e" % (rel(out["pc"][0], out["wave"][0]), rel(out["pc"][1], out["wave"][1])), flush=True)
