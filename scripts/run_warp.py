"""A few WARP epochs without torch: the target of the rocprofv3 passes behind the WARP roofline (DESIGN "WARP").
    python scripts/run_warp.py [shape=ml20m|c5] [epochs=4] [k=v backend knobs ...]
ml20m: bench.py's extra_warp workload (138,493 x 27,278, 20 M nnz, d=256, adagrad).  c5: BASELINE configs[4]'s shape on ONE GPU
(10 M x 1 M, 1 B nnz, d=256; 41.9 GB resident).  Prints one line per epoch: T (scored negatives per positive), accepted
fraction, kernel / aux / optimizer ms -- the per-epoch launches are matched to these by order in scripts/pmc_kernels.py."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from buffalo_amd import synth  # noqa: E402
from buffalo_amd.backend import CyWARP  # noqa: E402

modes = dict(kv.split("=") for kv in sys.argv[1:])
shape = modes.pop("shape", "ml20m")
epochs = int(modes.pop("epochs", 4))
out_path = modes.pop("out", "")
if shape == "c5":
    indptr, keys, P, Q, Qb = bench.warp_c5_inputs()
    U, I = P.shape[0], Q.shape[0]
else:
    csr = bench.load_matrix(shape, 7)
    U, I, indptr, keys = csr.num_users, csr.num_items, csr.indptr, csr.keys
    P, Q, Qb = synth.init_factors(U, I, bench.WARP_D, seed=7, signed=True)
    Qb *= 0
nnz, d = int(keys.shape[0]), P.shape[1]
g = CyWARP()
path = bench._opt_file(dict(bench.WARP_OPT, num_iters=epochs + 1))
assert g.init(path)
os.unlink(path)
g.sync_every_epoch = False
for k, v in modes.items():
    g.set_mode(k, int(v))
g.initialize_model(P, Q, Qb, nnz, True)
g.set_resident_csr(indptr, keys)
rows = []
for e in range(epochs):
    g.reset_stats()
    t0 = time.perf_counter()
    g.add_jobs(0, U, indptr, None)
    g.update_parameters()
    dt = time.perf_counter() - t0
    st = g.stats()
    rows.append(bench.warp_epoch_row(st, nnz, d, U, I, dt))
    print("run_warp", shape, "epoch", e, json.dumps(rows[-1]), flush=True)
if out_path:
    json.dump({"shape": shape, "U": U, "I": I, "nnz": nnz, "d": d, "epochs": rows}, open(out_path, "w"), indent=1)
