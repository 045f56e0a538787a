"""Per-rank cost of `bench.py --gpus N` measured on ONE GPU: EVERY rank's shard of the ML-20M-shaped matrix split N = 2, 4, 8 ways
(the epoch of an N-GPU run is its SLOWEST shard), each with a one-rank communicator attached (RCCL loaded, exchange kernels +
all-reduce of Q | Qb on the comm stream: the wire is a local copy) -- and rank 0's without, for what the machinery costs.  What is
left out is only the xGMI time of the 14 MB all-reduce, which the pipelined exchange puts behind the next walk.
    python scripts/shard_times.py [row_cost=<entries per row>] [<bfh_bpr_set_mode knob>=<int> ...]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from buffalo_amd import synth
from buffalo_amd.backend import Comm, CyBPR
from buffalo_amd.dist import shard_csr

csr = bench.load_matrix("ml20m", 7)
U, I, nnz = csr.num_users, csr.num_items, csr.nnz
comm = Comm(1, 0, Comm.unique_id(), 0)
out = {}
modes = dict(kv.split("=") for kv in sys.argv[1:])
row_cost = float(modes.pop("row_cost", 0.0))
P0, Q0, Qb0 = synth.init_factors(U, I, 128, seed=7)


def one(N, rank, with_comm, n=20):
    u0, u1, ip, keys, off = shard_csr(csr.indptr, csr.keys, rank, N, row_cost)
    P, Q, Qb = np.ascontiguousarray(P0[u0:u1]), Q0.copy(), Qb0.copy()
    obj = CyBPR()
    assert obj.init(bench.write_opt(bench.bpr_options(40)))
    obj.sync_every_epoch = False
    for k, v in modes.items():
        obj.set_mode(k, int(v))
    obj.initialize_model(P, Q, Qb, nnz, True)
    obj.set_cumulative_table(np.zeros(I, np.int64), I)
    obj.set_resident_csr(ip, keys)
    obj.set_shard(off, N)
    if with_comm:
        obj.set_comm(comm)
    for _ in range(4):
        obj.add_jobs(0, u1 - u0, ip, None)
        obj.update_parameters()
    obj.reset_stats()
    t0 = time.perf_counter()
    for _ in range(n):
        obj.add_jobs(0, u1 - u0, ip, None)
        obj.update_parameters()
    if with_comm:
        obj.comm_flush()
    dt = (time.perf_counter() - t0) / n
    st = obj.stats()
    r = {"epoch_ms": dt * 1e3, "kernel_ms": st["kernel_ms"] / n, "aux_ms": st["aux_ms"] / n, "launches": st["launches"] / n,
         "exchanges": st["exchanges"] / n, "local_triples": int(keys.shape[0]), "local_users": int(u1 - u0)}
    obj.set_comm(None)
    del obj
    return r


out["shards1_nocomm"] = one(1, 0, False)
out["shards1_comm"] = one(1, 0, True)
print("N=1", out["shards1_nocomm"], flush=True)
base = out["shards1_nocomm"]["epoch_ms"]
for N in (2, 4, 8):
    out["shards%d_rank0_nocomm" % N] = one(N, 0, False)
    rows = [one(N, r, True) for r in range(N)]
    out["shards%d_comm" % N] = rows
    ep = np.array([r["epoch_ms"] for r in rows])
    print("N=%d  per-rank epoch ms with the exchange machinery: %s" % (N, " ".join("%.3f" % e for e in ep)), flush=True)
    print("      users per rank %s   triples per rank %s" % ([r["local_users"] for r in rows], [r["local_triples"] for r in rows]))
    print("      max %.3f  min %.3f  mean %.3f  max/mean %.3f   rank 0 without communicator %.3f ms   -> %.2fx over 1 GPU (%.3f ms) from the SLOWEST "
          "shard, %.2fx from the mean, before xGMI time" % (ep.max(), ep.min(), ep.mean(), ep.max() / ep.mean(), out["shards%d_rank0_nocomm" % N]["epoch_ms"],
                                                               base / ep.max(), base, base / ep.mean()), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "shard_times.json"), "w"), indent=1)
