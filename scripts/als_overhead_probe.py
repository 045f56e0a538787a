"""Where does the wall time of an ALS epoch go?  Per-call wall times of precompute / partial_update with and without
torch imported first (bench.py's context)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "torch" in sys.argv[1:]:
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
import bench
from buffalo_amd import ingest, synth
from buffalo_amd.backend import CyALS
csr = bench.load_matrix("ml20m", 7)
U, I, nnz = csr.num_users, csr.num_items, csr.nnz
vals = (1 + np.random.default_rng(7).poisson(1.0, size=nnz)).astype(np.float32)
col = ingest.coo_to_csr(csr.keys, csr.rows(), vals, I, U)
P, Q, _ = synth.init_factors(U, I, 128, seed=7)
g = CyALS()
assert g.init(bench.write_opt(bench.ALS_OPT))
g.initialize_model(P, Q)
g.set_resident_csr(0, csr.indptr, csr.keys, vals)
g.set_resident_csr(1, col["indptr"], col["key"], col["val"])
g.set_mode("als_writeback", 0)
for e in range(6):
    ts = [time.perf_counter()]
    g.precompute(0); ts.append(time.perf_counter())
    g.partial_update(0, U, csr.indptr, None, None, 0); ts.append(time.perf_counter())
    g.precompute(1); ts.append(time.perf_counter())
    g.partial_update(0, I, col["indptr"], None, None, 1); ts.append(time.perf_counter())
    print("epoch", e, ["%.2f" % ((b - a) * 1e3) for a, b in zip(ts[:-1], ts[1:])], "total %.2f ms" % ((ts[-1] - ts[0]) * 1e3), g.stats()["kernel_ms"], flush=True)
