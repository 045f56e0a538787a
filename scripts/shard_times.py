"""Per-rank cost of `bench.py --gpus N` measured on ONE GPU: rank 0's shard of the ML-20M-shaped matrix split N ways, with
a one-rank communicator attached (RCCL loaded, exchange kernels + all-reduce of Q | Qb on the comm stream: the wire is a
local copy) and without.  What is left out is only the xGMI time of the 14 MB all-reduce, which the pipelined exchange puts
behind the next walk."""
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
for N in (1, 2, 4, 8):
    u0, u1, ip, keys, off = shard_csr(csr.indptr, csr.keys, 0, N)
    for with_comm in (False, True):
        P, Q, Qb = synth.init_factors(U, I, 128, seed=7)
        P = np.ascontiguousarray(P[u0:u1])
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
        for _ in range(5):
            obj.add_jobs(0, u1 - u0, ip, None)
            obj.update_parameters()
        obj.reset_stats()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            obj.add_jobs(0, u1 - u0, ip, None)
            obj.update_parameters()
        if with_comm:
            obj.comm_flush()
        dt = (time.perf_counter() - t0) / n
        st = obj.stats()
        name = "shards%d_%s" % (N, "comm" if with_comm else "nocomm")
        out[name] = {"epoch_ms": dt * 1e3, "kernel_ms": st["kernel_ms"] / n, "aux_ms": st["aux_ms"] / n, "launches": st["launches"] / n,
                     "exchanges": st["exchanges"] / n, "local_triples": int(keys.shape[0])}
        print(name, out[name], flush=True)
        obj.set_comm(None)
        del obj
base = out["shards1_nocomm"]["epoch_ms"]
for N in (2, 4, 8):
    print("N=%d: per-rank epoch %.3f ms with the exchange machinery -> %.2fx over 1 GPU (%.3f ms) before xGMI time"
          % (N, out["shards%d_comm" % N]["epoch_ms"], base / out["shards%d_comm" % N]["epoch_ms"], base))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "shard_times.json"), "w"), indent=1)
