"""Where does als_wide_kernel's time go at d = 160 (ML-20M shape)?  Row-kernel ms per half-epoch with the pass switched off ("als_debug" 16: timing only,
results are wrong) against the full kernel, split-f16 and fp32."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from buffalo_amd import ingest, synth
from buffalo_amd.backend import CyALS
d = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 160
csr = bench.load_matrix("ml20m", 7)
U, I, nnz = csr.num_users, csr.num_items, csr.nnz
vals = (1 + np.random.default_rng(7).poisson(1.0, size=nnz)).astype(np.float32)
col = ingest.coo_to_csr(csr.keys, csr.rows(), vals, I, U)
MAXT = {"als_wide_split_max_t": int(os.environ["ALS_WIDE_MAX_T"])} if "ALS_WIDE_MAX_T" in os.environ else {}   # e.g. 8: the split form at d = 256 too
CASES = (("split", dict(MAXT)), ("split, no pass", dict(MAXT, als_debug=16)), ("fp32", {"als_wide_split": 0}))
if "--split-only" in sys.argv:   # one line: the library in place (a variant build copied over it)
    CASES = (("split", {}),)
if "--grid" in sys.argv:         # "als_debug" 1024 (an experiment's host-side switch, not in the tree: profiles/r05_als_wide_d160.txt): two blocks per CU with the three-block binary
    CASES = (("split", {}), ("split, 2 blocks/CU", {"als_debug": 1024}), ("split, no pass", {"als_debug": 16}), ("split, no pass, 2 blocks/CU", {"als_debug": 1040}))
if "--study" in sys.argv:   # "als_debug" bits 32 / 64: consumers / producer reduced to the barriers (timing only)
    CASES = (("split", {}), ("split, no pass", {"als_debug": 16}), ("consumers idle", {"als_debug": 32}), ("producer idle", {"als_debug": 64}),
             ("both idle", {"als_debug": 96}))
for name, modes in CASES:
    P, Q, _ = synth.init_factors(U, I, d, seed=7)
    g = CyALS()
    assert g.init(bench.write_opt(dict(bench.ALS_OPT, d=d)))
    for k, v in modes.items():
        g.set_mode(k, v)
    g.initialize_model(P, Q)
    g.set_resident_csr(0, csr.indptr, csr.keys, vals)
    g.set_resident_csr(1, col["indptr"], col["key"], col["val"])
    g.set_mode("als_writeback", 0)
    out = []
    for axis, rows, ip in ((0, U, csr.indptr), (1, I, col["indptr"])):
        g.precompute(axis); g.partial_update(0, rows, ip, None, None, axis)
        g.reset_stats()
        for _ in range(3):
            g.precompute(axis); g.partial_update(0, rows, ip, None, None, axis)
        out.append(g.stats()["kernel_ms"] / 3)
    print("d=%d %-16s user half %.2f ms  item half %.2f ms  epoch %.2f ms" % (d, name, out[0], out[1], out[0] + out[1]), flush=True)
    del g
