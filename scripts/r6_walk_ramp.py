"""The headline kernel's time epoch by epoch from a cold start of the process (is the first handle's +6 % a clock ramp?); then 2 s of idling and again."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from buffalo_amd import synth
from buffalo_amd.backend import CyBPR
csr = bench.load_matrix("ml20m", 7)
U, I, nnz = csr.num_users, csr.num_items, csr.nnz
P, Q, Qb = synth.init_factors(U, I, bench.D, seed=7)
g = CyBPR()
assert g.init(bench.write_opt(bench.bpr_options(400)))
g.sync_every_epoch = False
g.initialize_model(P, Q, Qb, nnz, True)
g.set_cumulative_table(np.zeros(I, np.int64), I)
g.set_resident_csr(csr.indptr, csr.keys)
def series(n):
    out = []
    for _ in range(n):
        g.reset_stats()
        g.add_jobs(0, U, csr.indptr, None); g.update_parameters()
        st = g.stats()
        out.append(st["kernel_ms"] / st["launches"])
    return out
t0 = time.time()
s = series(150)
print("cold start, kernel ms per launch, epochs 1..150 (%.1f s):" % (time.time() - t0), " ".join("%.2f" % x for x in s), flush=True)
time.sleep(float(os.environ.get("IDLE", "3")))
s = series(60)
print("after idling:", " ".join("%.2f" % x for x in s), flush=True)
