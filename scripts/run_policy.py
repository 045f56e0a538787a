"""A few epochs of the bench workload (ML-20M-shaped, d=128, bench.py's options) under given backend knobs,
without torch: the target of the rocprofv3 passes.  python scripts/run_policy.py hogwild_atomic=3 [k=v ...] [epochs=N]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import bpr_options, load_matrix, write_opt  # noqa: E402
from buffalo_amd import synth  # noqa: E402
from buffalo_amd.backend import CyBPR  # noqa: E402

modes = dict(kv.split("=") for kv in sys.argv[1:])
epochs = int(modes.pop("epochs", 4))
lr = float(modes.pop("lr", 0.002))
shards = int(modes.pop("shards", 1))       # run on the first of `shards` nnz-balanced user shards (what one rank of N sees)
csr = load_matrix("ml20m", 7)
if shards > 1:
    from buffalo_amd.dist import shard_csr
    u0, u1, ip, keys, _ = shard_csr(csr.indptr, csr.keys, 0, shards)
    csr = synth.CSR(u1 - u0, csr.num_items, ip, keys, np.ones(keys.shape[0], np.float32))
U, I, nnz = csr.num_users, csr.num_items, csr.nnz
P, Q, Qb = synth.init_factors(U, I, 128, seed=7)
obj = CyBPR()
path = write_opt(bpr_options(epochs + 1, lr=lr))
assert obj.init(path)
os.unlink(path)
obj.sync_every_epoch = False
for k, v in modes.items():
    obj.set_mode(k, int(v))
obj.initialize_model(P, Q, Qb, nnz, True)
obj.set_cumulative_table(np.zeros(I, np.int64), I)
obj.set_resident_csr(csr.indptr, csr.keys)
obj.add_jobs(0, U, csr.indptr, None)
obj.update_parameters()
obj.reset_stats()
t0 = time.perf_counter()
for _ in range(epochs):
    obj.add_jobs(0, U, csr.indptr, None)
    obj.update_parameters()
dt = (time.perf_counter() - t0) / epochs
print("run_policy", modes, "epoch_ms %.3f" % (dt * 1e3), obj.stats(), flush=True)
