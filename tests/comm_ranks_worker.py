"""One rank of an N-process world that SHARES ONE GPU (test infrastructure for tests/test_comm_ranks_gpu.py).

RCCL refuses two ranks on one device, so on a one-GPU box the library's exchange code (csrc/sgd_base.hip exchange_*,
csrc/als_kernels.hpp publish_rows) runs with N > 1 over the shared-memory test transport of csrc/comm.hip
(BFH_COMM_TRANSPORT=shm): same Comm interface, same kernels, same call sequence as over RCCL -- only the wire differs.

    python tests/comm_ranks_worker.py SPEC.json      # spec: scenario, world, rank, uid (hex), out (npz path), knobs

No torch in here: the backend is ctypes over the C ABI, so eight of these start in a couple of seconds."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


# ---- the inputs every rank (and the single-process reference in the test) rebuilds from seeds ----------------------------------
def sgd_problem(spec):
    from conftest import bpr_opt, tiny_csr, warp_opt
    import helpers as H
    csr = tiny_csr(U=spec.get("U", 96), I=spec.get("I", 80), density=spec.get("density", 0.15), seed=spec.get("data_seed", 4))
    kind = spec["kind"]
    if kind == "warp":
        opt = warp_opt(d=spec.get("d", 64), random_seed=11, num_iters=spec["epochs"], lr=0.05, max_trials=30, threshold=0.5, reg_i=0.02)
    else:
        opt = bpr_opt(d=spec.get("d", 40), lr=spec.get("lr", 0.05), min_lr=spec.get("min_lr", 0.01), num_iters=spec["epochs"], random_seed=5,
                      **spec.get("opt", {}))
    d = opt["d"]
    vdim = ((d + 31) // 32) * 32
    rng = np.random.default_rng(1)
    P = H.pad(rng.normal(scale=0.3, size=(csr.num_users, d)).astype(np.float32), vdim)
    Q = H.pad(rng.normal(scale=0.3, size=(csr.num_items, d)).astype(np.float32), vdim)
    Qb = rng.normal(scale=0.1, size=(csr.num_items, 1)).astype(np.float32)
    if kind == "warp":
        Qb[:] = 0
    return csr, opt, P, Q, Qb


def make_sgd(spec, csr, opt, P, Q, Qb, rank, world, comm):
    """The handle of one rank: its user shard of P and of the CSR, Q / Qb replicated (bench.py's set-up)."""
    import buffalo_amd.backend as B
    import helpers as H
    from buffalo_amd.dist import shard_csr
    cls = B.CyWARP if spec["kind"] == "warp" else B.CyBPR
    u0, u1, ip, keys, off = shard_csr(csr.indptr, csr.keys, rank, world)
    if spec.get("empty_rank") == rank:      # this rank owns no rows at all: every call of it is an empty chunk
        pass
    Pl = np.ascontiguousarray(P[u0:u1])
    obj = cls()
    assert obj.init(H.write_opt(dict(opt, accelerator=True)))
    for k, v in spec.get("modes", {}).items():
        obj.set_mode(k, v)
    obj.initialize_model(Pl, Q, Qb, csr.nnz, True)
    obj.set_cumulative_table(H.cum_table(csr, opt), csr.num_items)
    obj.set_resident_csr(ip, keys)
    obj.set_shard(off, world)
    if comm is not None:
        obj.set_comm(comm)
    return obj, Pl, (u0, u1, ip)


def chunk_edges(n_rows, n_chunks, ragged=False):
    e = np.linspace(0, n_rows, n_chunks + 1).astype(int)
    if ragged and n_chunks >= 2:
        e[1] = e[0]          # the first chunk of the call list is EMPTY on this rank
    return e


def run_sgd_rank(spec, rank, world, comm):
    csr, opt, P, Q, Qb = sgd_problem(spec)
    Q, Qb = Q.copy(), Qb.copy()
    obj, Pl, (u0, u1, ip) = make_sgd(spec, csr, opt, P, Q, Qb, rank, world, comm)
    edges = chunk_edges(u1 - u0, spec.get("chunks", 1), ragged=spec.get("ragged_rank") == rank)
    for _ in range(spec["epochs"]):
        for a, b in zip(edges[:-1], edges[1:]):
            obj.add_jobs(int(a), int(b), ip, None)
        obj.update_parameters()
    if comm is not None:
        obj.comm_flush()
    st = obj.stats()
    obj.synchronize(True)
    obj.set_comm(None)
    return dict(P=Pl, Q=Q, Qb=Qb, u0=u0, u1=u1, exchanges=st["exchanges"])


def als_problem(spec):
    from conftest import als_opt, tiny_csr
    import helpers as H
    csr = tiny_csr(U=spec.get("U", 150), I=spec.get("I", 110), density=0.12, seed=9, counts=True)
    opt = als_opt(d=spec.get("d", 32), alpha=4.0, reg_u=0.2, reg_i=0.3, num_iters=spec["epochs"], compute_loss_on_training=True,
                  optimizer=spec.get("optimizer", "manual_cg"))
    d = opt["d"]
    vdim = ((d + 31) // 32) * 32
    rng = np.random.default_rng(3)
    P = H.pad(np.abs(rng.normal(scale=0.1, size=(csr.num_users, d))).astype(np.float32), vdim)
    Q = H.pad(np.abs(rng.normal(scale=0.1, size=(csr.num_items, d))).astype(np.float32), vdim)
    return csr, opt, P, Q


def run_als_rank(spec, rank, world, comm):
    import helpers as H
    from buffalo_amd.backend import CyALS
    from buffalo_amd.dist import CommDataParallelALS
    csr, opt, P, Q = als_problem(spec)
    t = csr.transpose()
    obj = CyALS()
    assert obj.init(H.write_opt(dict(opt, accelerator=True)))
    obj.initialize_model(P, Q)
    obj.set_resident_csr(0, csr.indptr, csr.keys, csr.vals)
    obj.set_resident_csr(1, t.indptr, t.keys, t.vals)
    losses = []
    if comm is not None:
        obj.set_comm(comm)
        dp = CommDataParallelALS(obj, comm, (csr.indptr, t.indptr), csr.num_users, csr.num_items)
        for _ in range(spec["epochs"]):
            losses.append(dp.epoch())
    else:
        obj.set_mode("als_writeback", 0)
        for _ in range(spec["epochs"]):
            tot = np.zeros(2)
            for axis, mat in ((0, csr), (1, t)):
                obj.precompute(axis)
                tot += obj.partial_update(0, mat.num_users, mat.indptr, None, None, axis)
            losses.append(tuple(tot))
    obj.synchronize(True)
    if comm is not None:
        obj.set_comm(None)
    return dict(P=P, Q=Q, losses=np.asarray(losses, dtype=np.float64))


def main():
    spec = json.load(open(sys.argv[1]))
    from buffalo_amd.backend import Comm
    comm = Comm(spec["world"], spec["rank"], bytes.fromhex(spec["uid"]), 0)
    comm.self_test()
    assert comm.all_reduce([float(spec["rank"] + 1)])[0] == spec["world"] * (spec["world"] + 1) / 2
    fn = {"sgd": run_sgd_rank, "als": run_als_rank}[spec["scenario"]]
    out = fn(spec, spec["rank"], spec["world"], comm)
    np.savez(spec["out"], **out)
    del comm


if __name__ == "__main__":
    main()
