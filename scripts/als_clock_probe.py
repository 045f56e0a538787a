"""The shader clock the config-#3 ALS row kernel (als_pc_kernel) sees, with and without its matrix instructions (als_debug bit 1024: s_memtime ticks per
100 MHz s_memrealtime tick over the kernel, workgroup 0; bits 16 / 1 / 32 switch the matrix instructions / the block solve / the preparation arithmetic off --
timing only, results are wrong with them).  Round 6 (profiles/r06_als_ts_first_contact.txt): 2.30-2.40 GHz without the matrix instructions, 1.93-2.10 GHz with
them -- a matrix pipe that is busy a quarter of the time costs 12-16 % of clock, which is what the "cost" of the hidden matrix instructions is.
    python scripts/als_clock_probe.py"""
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


def timing(modes, epochs=4):
    P, Q, _ = synth.init_factors(U, I, D, seed=7)
    g = CyALS()
    path = bench._opt_file(OPT)
    assert g.init(path)
    os.unlink(path)
    g.initialize_model(P, Q)
    g.set_resident_csr(0, csr.indptr, csr.keys, vals)
    g.set_resident_csr(1, col["indptr"], col["key"], col["val"])
    for k, v in dict(modes, als_writeback=0).items():
        g.set_mode(k, v)
    per = {0: [], 1: []}
    for ep in range(epochs):
        for axis in (0, 1):
            rows, ip = (U, csr.indptr) if axis == 0 else (I, col["indptr"])
            g.precompute(axis)
            g.reset_stats()
            g.partial_update(0, rows, ip, None, None, axis)
            if ep:
                per[axis].append(g.stats()["kernel_ms"])
    print("%-28s user half-epoch %.3f ms  item half-epoch %.3f ms  sum %.3f  shader clock of workgroup 0 over the last launch %d MHz"
          % (modes, np.mean(per[0]), np.mean(per[1]), np.mean(per[0]) + np.mean(per[1]), g.device_buffer("als_pc_clock_mhz")[1]), flush=True)
    del g


for bits, what in ((0, "the kernel"), (16, "no matrix instructions"), (17, "... and no block solve"), (49, "... and no preparation arithmetic")):
    print(what + ":", end=" ")
    timing({"als_debug": 1024 + bits})
