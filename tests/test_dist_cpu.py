"""Multi-process (world_size 2, gloo, CPU) tests of the data-parallel layer (buffalo_amd/dist.py).

The local engine is the CPU oracle in deterministic mode, so these tests check the *distribution*
logic the HIP engine relies on: nnz-balanced user sharding, shard offsets that keep the counter
sampler identical to the single-process run, and the delta all-reduce of the replicated tensors."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleEngine:
    """DataParallelSGD engine backed by the oracle (tests only)."""

    def __init__(self, o, Q, Qb):
        self.o, self.Q, self.Qb = o, Q, Qb

    def replicated_tensors(self, kind):
        import torch
        if kind == "model":
            return [torch.from_numpy(self.Q), torch.from_numpy(self.Qb)]
        return [torch.from_numpy(self.o.state_view("gradQ")), torch.from_numpy(self.o.state_view("gradQb"))]

    def add_jobs(self, a, b, indptr, keys):
        return self.o.add_jobs(a, b, indptr, keys)

    def wait(self):
        pass

    def update_parameters(self):
        self.o.update_parameters()


def _make_problem(kind):
    from conftest import bpr_opt, tiny_csr, warp_opt
    csr = tiny_csr(U=60, I=50, density=0.15, seed=3)
    d = 12
    rng = np.random.default_rng(5)
    P = rng.normal(scale=0.3, size=(60, d)).astype(np.float32)
    Q = rng.normal(scale=0.3, size=(50, d)).astype(np.float32)
    Qb = rng.normal(scale=0.1, size=(50, 1)).astype(np.float32)
    if kind == "warp":
        opt = warp_opt(d=d, random_seed=9, max_trials=10, threshold=0.3, num_iters=2, lr=0.05)
        Qb *= 0
    elif kind == "bpr_adagrad":
        opt = bpr_opt(d=d, random_seed=9, optimizer="adagrad", lr=0.05, num_iters=2)
    else:
        opt = bpr_opt(d=d, random_seed=9, optimizer="sgd", lr=0.05, min_lr=0.01, num_iters=2)
    return csr, opt, P, Q, Qb


def _oracle(kind, opt, P, Q, Qb, nnz_total):
    import helpers as H
    from oracle import oracle as orc
    o = (orc.OracleWARP if kind == "warp" else orc.OracleBPRMF)()
    assert o.init(H.write_opt(opt))
    o.initialize_model(P, Q, Qb, nnz_total)
    o.set_cumulative_table(np.zeros(Q.shape[0], np.int64), Q.shape[0])
    o.set_modes(sampler="counter", pos_order="csr", inline=True)
    o.launch_workers()
    return o


def _worker(rank, world, port, kind, out_dir):
    import torch.distributed as dist
    from dist_harness import DataParallelSGD, shard_csr
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    csr, opt, P, Q, Qb = _make_problem(kind)
    u0, u1, ip, keys, off = shard_csr(csr.indptr, csr.keys, rank, world)
    Pl = np.ascontiguousarray(P[u0:u1])
    o = _oracle(kind, opt, Pl, Q, Qb, csr.nnz)
    o.set_shard(off, world)
    dp = DataParallelSGD(OracleEngine(o, Q, Qb), opt["optimizer"])
    for _ in range(2):
        n = u1 - u0
        for a, b in ((0, n // 2), (n // 2, n)):      # two minibatches per epoch
            beg = 0 if a == 0 else int(ip[a - 1])
            end = int(ip[b - 1])
            dp.minibatch(a, b, ip, np.ascontiguousarray(keys[beg:end]))
        dp.end_epoch()
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), P=Pl, Q=Q, Qb=Qb, u0=u0, u1=u1)
    dist.barrier()
    dist.destroy_process_group()


def _run_world(kind, tmp_path, world=2):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, kind, str(tmp_path)), nprocs=world, join=True)
    return [np.load(os.path.join(str(tmp_path), "r%d.npz" % r)) for r in range(world)]


def test_shard_bounds_balance_nnz():
    from buffalo_amd import synth
    from buffalo_amd.dist import shard_bounds, shard_csr
    csr = synth.generate(2000, 500, 40000, seed=1)
    for world in (1, 2, 3, 8):
        b = shard_bounds(csr.indptr, world)
        assert b[0] == 0 and b[-1] == 2000 and all(x <= y for x, y in zip(b, b[1:]))
        sizes, total = [], 0
        for r in range(world):
            u0, u1, ip, keys, off = shard_csr(csr.indptr, csr.keys, r, world)
            assert off == total and (len(ip) == 0 or ip[-1] == keys.shape[0])
            total += keys.shape[0]
            sizes.append(keys.shape[0])
        assert total == csr.nnz
        assert max(sizes) - min(sizes) <= 2 * int(np.diff(np.concatenate([[0], csr.indptr])).max())


@pytest.mark.parametrize("kind", ["bpr_adagrad", "warp"])
def test_gradient_allreduce_equals_single_process(kind, tmp_path):
    """adam/adagrad/WARP: P,Q frozen inside an epoch => 2 ranks == 1 process up to summation order."""
    import helpers as H
    outs = _run_world(kind, tmp_path)
    csr, opt, P, Q, Qb = _make_problem(kind)
    o = _oracle(kind, opt, P, Q, Qb, csr.nnz)
    for _ in range(2):
        o.add_jobs(0, csr.num_users, csr.indptr, csr.keys)
        o.update_parameters()
    Pm = np.concatenate([z["P"] for z in outs])
    assert H.relerr(Pm, P) < 1e-5
    for z in outs:                                   # replicas stay identical and match
        assert H.relerr(z["Q"], Q) < 1e-5 and H.relerr(z["Qb"], Qb) < 1e-5 + (kind == "warp")
    np.testing.assert_array_equal(outs[0]["Q"], outs[1]["Q"])


def test_sgd_delta_allreduce_keeps_replicas_consistent(tmp_path):
    """Hogwild sgd across ranks is local-SGD with summed item deltas: replicas must be identical
    after every exchange, every rank must have moved Q, and the result stays near the sequential one."""
    import helpers as H
    outs = _run_world("bpr_sgd", tmp_path)
    csr, opt, P, Q, Qb = _make_problem("bpr_sgd")
    Q0 = Q.copy()
    o = _oracle("bpr_sgd", opt, P, Q, Qb, csr.nnz)
    for _ in range(2):
        o.add_jobs(0, csr.num_users, csr.indptr, csr.keys)
        o.update_parameters()
    np.testing.assert_array_equal(outs[0]["Q"], outs[1]["Q"])
    np.testing.assert_array_equal(outs[0]["Qb"], outs[1]["Qb"])
    assert not np.array_equal(outs[0]["Q"], Q0)
    # same negatives are drawn (shard offsets), only the visibility of the other rank's item updates differs
    assert H.relerr(outs[0]["Q"], Q) < 0.05
    assert H.relerr(np.concatenate([z["P"] for z in outs]), P) < 0.05


# ------------------------------------------------------------------------------------------------
# The pipelined exchange (what a handle does by itself once bfh_*_set_comm attached an RCCL rank)
# ------------------------------------------------------------------------------------------------
def _pipe_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from dist_harness import DataParallelSGD, shard_csr
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    csr, opt, P, Q, Qb = _make_problem("bpr_sgd")
    u0, u1, ip, keys, off = shard_csr(csr.indptr, csr.keys, rank, world)
    Pl = np.ascontiguousarray(P[u0:u1])
    o = _oracle("bpr_sgd", opt, Pl, Q, Qb, csr.nnz)
    o.set_shard(off, world)
    dp = DataParallelSGD(OracleEngine(o, Q, Qb), "sgd", pipelined=True)
    own = np.zeros_like(Q, dtype=np.float64)             # what THIS rank's walks added to Q, summed over the run
    stale = []
    for _ in range(3):
        n = u1 - u0
        for a, b in ((0, n // 3), (n // 3, n)):
            beg = 0 if a == 0 else int(ip[a - 1])
            end = int(ip[b - 1])
            before = Q.copy()
            o.add_jobs(a, b, ip, np.ascontiguousarray(keys[beg:end]))
            own += Q.astype(np.float64) - before
            dp.pipe.begin()                                # publishes this walk's delta; the previous one lands first
            stale.append(dp.pipe.work is not None)
        dp.end_epoch()
    dp.flush()
    np.savez(os.path.join(out_dir, "p%d.npz" % rank), Q=Q, Qb=Qb, own=own, in_flight=np.array(stale))
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_exchange_applies_every_delta_once(tmp_path):
    """One exchange in flight behind the next walk: after the final flush the replicas are identical and equal
    Q0 + the sum over ranks of everything their own walks added -- nothing lost, nothing applied twice."""
    import torch.multiprocessing as mp
    import helpers as H
    port = _free_port()
    mp.spawn(_pipe_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    outs = [np.load(os.path.join(str(tmp_path), "p%d.npz" % r)) for r in range(2)]
    _, _, _, Q0, _ = _make_problem("bpr_sgd")
    np.testing.assert_array_equal(outs[0]["Q"], outs[1]["Q"])
    np.testing.assert_array_equal(outs[0]["Qb"], outs[1]["Qb"])
    want = Q0.astype(np.float64) + outs[0]["own"] + outs[1]["own"]
    assert H.relerr(outs[0]["Q"], want) < 1e-5
    assert all(z["in_flight"].all() for z in outs)       # every walk left its exchange in flight


def _quality_worker(rank, world, port, pipelined, seed, out_dir, minibatches=1):
    import torch.distributed as dist
    import helpers as H
    from buffalo_amd import synth
    from dist_harness import DataParallelSGD, shard_csr
    from conftest import bpr_opt
    from oracle import oracle as orc
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    csr, vali = synth.planted(600, 400, d_true=6, density=0.06, seed=7)
    d, epochs = 16, 30
    opt = bpr_opt(d=d, lr=0.05, min_lr=0.01, num_iters=epochs, random_seed=seed, reg_u=0.01, reg_i=0.01, reg_j=0.01, reg_b=0.01)
    P, Q, Qb = synth.init_factors(600, 400, d, seed=7)
    u0, u1, ip, keys, off = shard_csr(csr.indptr, csr.keys, rank, world)
    Pl = np.ascontiguousarray(P[u0:u1])
    o = orc.OracleBPRMF()
    assert o.init(H.write_opt(opt))
    o.initialize_model(Pl, Q, Qb, csr.nnz)
    o.set_cumulative_table(np.zeros(400, np.int64), 400)
    o.set_modes(sampler="counter", pos_order="csr", inline=True)
    o.set_shard(off, world)
    o.launch_workers()
    dp = DataParallelSGD(OracleEngine(o, Q, Qb), "sgd", pipelined=pipelined) if world > 1 else None
    for _ in range(epochs):
        if dp is not None:
            n = u1 - u0
            for x in range(minibatches):
                a, b = n * x // minibatches, n * (x + 1) // minibatches
                kb, ke = (0 if a == 0 else int(ip[a - 1])), int(ip[b - 1])
                dp.minibatch(a, b, ip, np.ascontiguousarray(keys[kb:ke]))
            dp.end_epoch()
        else:
            o.add_jobs(0, u1 - u0, ip, keys)
            o.update_parameters()
    if dp is not None:
        dp.flush()
    np.savez(os.path.join(out_dir, "q%d_%d_%d_%d.npz" % (world, int(pipelined), seed, rank)), P=Pl, Q=Q, Qb=Qb, u0=u0, u1=u1)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_local_sgd_quality_matches_single_process(tmp_path):
    """Hogwild sgd across ranks = local SGD with summed item deltas.  NDCG@10 on planted data against the single-process
    run (mirrors the reference's threshold tests, tests/algo/test_bpr.py:38-47; same 25 % band as the single-GPU Hogwild
    test): blocking exchange once per epoch, and the pipelined exchange the library applies (one exchange in flight behind
    the next walk) at four exchange points per epoch.  This 400-item problem at lr 0.05 is the worst case -- every row is
    touched by every rank in every interval; scripts/local_sgd_study.py has the same comparison at BASELINE scale and
    profiles/r02_local_sgd_toy_sweep.txt the sweep over ranks x exchange points (1 process 0.275; 8 ranks blocking 1x
    0.208, pipelined 1x 0.12, 4x 0.21, 8x 0.23: a delayed exchange needs ~4 exchange points per epoch to match a blocking
    one, which is why a handle with a communicator cuts a call into lr-dependent exchange segments)."""
    import torch.multiprocessing as mp
    import helpers as H
    from buffalo_amd import synth
    csr, vali = synth.planted(600, 400, d_true=6, density=0.06, seed=7)

    def ndcg(world, pipelined, seed, minibatches=1):
        if world == 1:
            _quality_worker(0, 1, 0, False, seed, str(tmp_path))
        else:
            mp.spawn(_quality_worker, args=(world, _free_port(), pipelined, seed, str(tmp_path), minibatches), nprocs=world, join=True)
        zs = [np.load(os.path.join(str(tmp_path), "q%d_%d_%d_%d.npz" % (world, int(pipelined), seed, r))) for r in range(world)]
        P = np.concatenate([z["P"] for z in zs])
        return H.ndcg_at_k(P, zs[0]["Q"], csr, vali, Qb=zs[0]["Qb"])

    single = [ndcg(1, False, s) for s in (7, 8, 9)]
    P0, Q0, Qb0 = synth.init_factors(600, 400, 16, seed=7)
    base = H.ndcg_at_k(P0, Q0, csr, vali, Qb=Qb0)
    mean, spread = float(np.mean(single)), float(np.max(single) - np.min(single))
    assert mean > 3 * max(base, 0.01)
    for pipelined, mb in ((False, 1), (True, 4)):
        got = ndcg(2, pipelined, 7, mb)
        print("local-SGD NDCG@10: 1 process %s (spread %.4f), 2 ranks %s, %d exchange(s) per epoch: %.4f"
              % (["%.4f" % x for x in single], spread, "pipelined" if pipelined else "blocking", mb, got))
        assert got > 3 * max(base, 0.01)
        assert abs(got - mean) <= 0.25 * mean, (got, single)


# ------------------------------------------------------------------------------------------------
# ALS: row shards + broadcast of the solved rows
# ------------------------------------------------------------------------------------------------
class OracleAlsEngine:
    """DataParallelALS engine backed by the oracle (tests only)."""

    def __init__(self, o, P, Q, csr, t):
        self.o, self.F, self.mats = o, (P, Q), (csr, t)

    def precompute(self, axis):
        self.o.precompute(axis)

    def partial_update(self, a, b, axis):
        import helpers as H
        m = self.mats[axis]
        if a == b:
            return 0.0, 0.0
        keys, vals = H.chunk_arrays(m, a, b)
        return tuple(self.o.partial_update(a, b, m.indptr, keys, vals, axis))

    def factor_tensor(self, axis):
        import torch
        return torch.from_numpy(self.F[axis])

    def wait(self):
        pass


def _als_problem(optimizer, d):
    from conftest import als_opt, tiny_csr
    csr = tiny_csr(U=70, I=45, density=0.2, seed=11, counts=True)
    rng = np.random.default_rng(2)
    P = np.abs(rng.normal(scale=0.2, size=(70, d))).astype(np.float32)
    Q = np.abs(rng.normal(scale=0.2, size=(45, d))).astype(np.float32)
    opt = als_opt(d=d, optimizer=optimizer, num_iters=2, alpha=4.0, reg_u=0.1, reg_i=0.2, block_size=8)
    return csr, opt, P, Q


def _als_oracle(opt, P, Q):
    import helpers as H
    from oracle import oracle as orc
    o = orc.OracleALS()
    assert o.init(H.write_opt(opt))
    o.initialize_model(P, Q)
    return o


def _als_worker(rank, world, port, optimizer, d, out_dir):
    import torch.distributed as dist
    from dist_harness import DataParallelALS
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    csr, opt, P, Q = _als_problem(optimizer, d)
    t = csr.transpose()
    dp = DataParallelALS(OracleAlsEngine(_als_oracle(opt, P, Q), P, Q, csr, t), (csr.indptr, t.indptr))
    losses = [dp.epoch() for _ in range(2)]
    np.savez(os.path.join(out_dir, "als%d.npz" % rank), P=P, Q=Q, losses=np.array(losses))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("optimizer,d", [("manual_cg", 12), ("ialspp", 20)])
def test_als_row_shards_equal_single_process(optimizer, d, tmp_path):
    """Every row is solved by exactly one rank from identical inputs => the 2-rank factors equal the
    1-process factors bit for bit; the (nume, deno) loss pairs agree to summation order."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_als_worker, args=(2, port, optimizer, d, str(tmp_path)), nprocs=2, join=True)
    outs = [np.load(os.path.join(str(tmp_path), "als%d.npz" % r)) for r in range(2)]
    csr, opt, P, Q = _als_problem(optimizer, d)
    t = csr.transpose()
    eng = OracleAlsEngine(_als_oracle(opt, P, Q), P, Q, csr, t)
    losses = []
    for _ in range(2):
        tot = np.zeros(2)
        for axis, m in ((0, csr), (1, t)):
            eng.precompute(axis)
            tot += eng.partial_update(0, m.num_users, axis)
        losses.append(tot)
    for z in outs:
        np.testing.assert_array_equal(z["P"], P)
        np.testing.assert_array_equal(z["Q"], Q)
        np.testing.assert_allclose(z["losses"], np.array(losses), rtol=1e-9)
