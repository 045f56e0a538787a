"""A/B of the config-#3 ALS row kernel, round 6: the producer / consumer pairs (als_pc_kernel) against the tile split (als_ts_kernel, "als_ts" = 1):
kernel time per half-epoch, ablations (als_debug bits: 1 no block solve, 16 no matrix instructions, 32 no preparation arithmetic), and the two results
from ONE warm state (same factors, same FF: the difference is the kernels' own).
    python scripts/als_ts_ab.py [--ablate] [--timing-only]"""
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
D = bench.D
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
    out = g.partial_update(0, rows, ip, None, None, axis)
    return g.stats()["kernel_ms"], out


def timing(modes, epochs=4):
    P, Q, _ = synth.init_factors(U, I, D, seed=7)
    g = make(P, Q, dict(modes, als_writeback=0))
    per = {0: [], 1: []}
    for ep in range(epochs):
        for axis in (0, 1):
            ms, _ = half(g, axis)
            if ep:
                per[axis].append(ms)
    clk = g.device_buffer("als_pc_clock_mhz")[1] if modes.get("als_debug", 0) & 1024 else 0
    print("%-44s user half-epoch %.3f ms  item half-epoch %.3f ms  sum %.3f  (one pair per SIMD in %d of 256 workgroups%s)"
          % (modes, np.mean(per[0]), np.mean(per[1]), np.mean(per[0]) + np.mean(per[1]), g.device_buffer("als_pc_same_simd")[1],
             "; shader clock of workgroup 0 over the last launch %d MHz" % clk if clk else ""), flush=True)
    del g


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


for m in ({"als_ts": 0}, {"als_ts": 1}, {"als_ts": 0}, {"als_ts": 1}):
    timing(m)
if "--ablate" in sys.argv:   # results are wrong with these, timings only
    for bits, what in ((1, "no block solve"), (16, "no matrix instructions"), (17, "neither"), (17 + 32, "neither, no preparation arithmetic"), (32, "no preparation arithmetic")):
        for ts in (0, 1):
            print("als_debug %d (%s):" % (bits, what), end=" ")
            timing({"als_ts": ts, "als_debug": bits}, epochs=3)
if "--clock" in sys.argv:   # als_debug bit 1024: the shader clock the kernel sees, with and without its matrix instructions
    for ts in (0, 1):
        for bits in (1024, 1024 + 16, 1024 + 17, 1024 + 49):
            timing({"als_ts": ts, "als_debug": bits}, epochs=3)
if "--timing-only" in sys.argv:
    sys.exit(0)
# the same warm state through both kernels (compute_loss on: the two loss formulations side by side)
res = {}
for ts in (0, 1):
    P, Q, _ = synth.init_factors(U, I, D, seed=7)
    g = make(P, Q, {"als_ts": 0})
    for ep in range(2):
        half(g, 0), half(g, 1)
    g.set_mode("als_ts", ts)
    _, lu = half(g, 0)
    g.synchronize(True)
    Pu = P.copy()
    _, li = half(g, 1)
    g.synchronize(True)
    res[ts] = (Pu, Q.copy(), lu, li)
    del g
print("third epoch from one warm state: user half max|ts - pairs| / max|pairs| = %.3e, item half %.3e" % (rel(res[1][0], res[0][0]), rel(res[1][1], res[0][1])))
print("loss (nume, deno): pairs user %s item %s | ts user %s item %s" % (res[0][2], res[0][3], res[1][2], res[1][3]))
