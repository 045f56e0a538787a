"""N = 1: the walk with two merge segments per epoch (default), with one (xcd_sync_updates = 2^25) and with a one-rank communicator attached (one segment + the exchange
machinery), handle after handle in one process.  Why: scripts/shard_times.py showed the communicator case 11 % faster per epoch than the plain one."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from buffalo_amd import synth
from buffalo_amd.backend import Comm, CyBPR
csr = bench.load_matrix("ml20m", 7)
U, I, nnz = csr.num_users, csr.num_items, csr.nnz
comm = Comm(1, 0, Comm.unique_id(), 0)
P0, Q0, Qb0 = synth.init_factors(U, I, 128, seed=7)


def one(label, modes, with_comm, shard, n=20):
    P, Q, Qb = P0.copy(), Q0.copy(), Qb0.copy()
    obj = CyBPR()
    assert obj.init(bench.write_opt(bench.bpr_options(40)))
    obj.sync_every_epoch = False
    for k, v in modes.items():
        obj.set_mode(k, int(v))
    obj.initialize_model(P, Q, Qb, nnz, True)
    obj.set_cumulative_table(np.zeros(I, np.int64), I)
    obj.set_resident_csr(csr.indptr, csr.keys)
    if shard:
        obj.set_shard(0, 1)
    if with_comm:
        obj.set_comm(comm)
    for _ in range(4):
        obj.add_jobs(0, U, csr.indptr, None); obj.update_parameters()
    obj.reset_stats()
    for _ in range(n):
        obj.add_jobs(0, U, csr.indptr, None); obj.update_parameters()
    if with_comm:
        obj.comm_flush()
    st = obj.stats()
    print("%-44s kernel %.3f ms per epoch in %.1f launches, aux %.3f, merges %.1f, exchanges %.1f" % (label, st["kernel_ms"] / n, st["launches"] / n, st["aux_ms"] / n,
          st["merges"] / n, st["exchanges"] / n), flush=True)
    if with_comm:
        obj.set_comm(None)
    del obj


for rep in range(2):
    one("default (two segments)", {}, False, False)
    one("xcd_sync_updates = 2^25 (one segment)", {"xcd_sync_updates": 1 << 25}, False, False)
    one("set_shard(0, 1) only", {}, False, True)
    one("one-rank communicator", {}, True, True)
    one("one-rank communicator, comm_segments = 2", {"comm_segments": 2}, True, True)
