"""Where the config-#3 ALS epoch goes: the row kernel's time per half-epoch with pieces switched off ("als_debug" bits -- timing only,
the results of such runs are wrong): 1 no in-register block solve, 2 no FF tiles / FF p0 before the pass, 4 no per-entry residual dot.
    python scripts/als_ablation.py [bits ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from buffalo_amd import ingest, synth
from buffalo_amd.backend import CyALS

csr = bench.load_matrix("ml20m", 7)
U, I, nnz = csr.num_users, csr.num_items, csr.nnz
vals = (1 + np.random.default_rng(7).poisson(1.0, size=nnz)).astype(np.float32)
col = ingest.coo_to_csr(csr.keys, csr.rows(), vals, I, U)
for bits in [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 7]:
    P, Q, _ = synth.init_factors(U, I, bench.D, seed=7)
    g = CyALS()
    path = bench._opt_file(bench.ALS_OPT)
    assert g.init(path)
    os.unlink(path)
    g.initialize_model(P, Q)
    g.set_resident_csr(0, csr.indptr, csr.keys, vals)
    g.set_resident_csr(1, col["indptr"], col["key"], col["val"])
    g.set_mode("als_writeback", 0)
    per = {0: [], 1: []}
    for ep in range(5):
        for axis, rows, ip in ((0, U, csr.indptr), (1, I, col["indptr"])):
            g.set_mode("als_debug", bits if ep else 0)      # a clean first epoch: the timed ones start from a sane model
            g.precompute(axis)
            g.reset_stats()
            g.partial_update(0, rows, ip, None, None, axis)
            if ep:
                per[axis].append(g.stats()["kernel_ms"])
    print("als_debug=%d  user half-epoch %.3f ms  item half-epoch %.3f ms  sum %.3f" % (bits, np.mean(per[0]), np.mean(per[1]), np.mean(per[0]) + np.mean(per[1])), flush=True)
    del g
