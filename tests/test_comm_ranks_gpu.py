"""The library's exchange code with N > 1 ranks, on ONE GPU: N processes (tests/comm_ranks_worker.py) share the device and talk
over the shared-memory test transport of csrc/comm.hip (BFH_COMM_TRANSPORT=shm; RCCL refuses two ranks on one device).
Everything above the wire is the product path: `bfh_*_set_comm`, exchange_arm / _begin / _finish / exchange_weight_kernel,
exchange_gradients inside update_parameters, `bfh_als_publish_rows`.

What is held to what:
  * adagrad / adam / WARP (frozen-epoch paths): the N-rank model equals the single-GPU model up to fp32 summation order
    (the sample stream is keyed by the entry's position in the WHOLE matrix, so the ranks draw what one GPU draws);
  * sgd, deterministic walk: bit-identical to the protocol restated in numpy over N no-comm handles run one after the
    other in this process (plain sum), 1e-5 with the saturation weights restated in float64;
  * after a flush the replicas of Q / Qb are bit-identical on every rank -- also for the Hogwild item-major walk and the
    pipelined (comm_segments > 1) exchange, and when a rank's chunk list contains empty chunks;
  * ALS: publish_rows leaves every rank with the single-GPU factors, bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import comm_ranks_worker as W
import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_world(tmp_path, world, spec, timeout=240, order=None):
    """`order="ring"`: the test transport sums every segment of a buffer in the rotated order a ring all-reduce produces
    (csrc/comm.hip shm_all_reduce) instead of in rank order."""
    os.makedirs(str(tmp_path), exist_ok=True)
    # the ranks load libbuffalo_hip_test.so (BFH_LIBRARY=test): the product library this process has loaded does not contain the test
    # transport (csrc/comm_test_transport.hpp, -DBFH_TEST_TRANSPORT) and refuses the knob
    env = dict(os.environ, BFH_COMM_TRANSPORT="shm", BFH_LIBRARY="test")
    if order:
        env["BFH_COMM_SHM_ORDER"] = order
    uid = (b"BFHSHM1\x00" + os.urandom(16)).ljust(128, b"\x00").hex()   # what bfh_comm_unique_id makes there: magic + 16 random bytes
    procs, outs = [], []
    for r in range(world):
        out = str(tmp_path / ("rank%d.npz" % r))
        path = str(tmp_path / ("spec%d.json" % r))
        with open(path, "w") as f:
            json.dump(dict(spec, world=world, rank=r, uid=uid, out=out), f)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "comm_ranks_worker.py"), path], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs.append(out)
    logs, failed = [], False
    for p in procs:
        try:
            log, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            log, _ = p.communicate()
            log += "\n[timeout]"
        logs.append(log)
        failed |= p.returncode != 0
    assert not failed, "\n-----\n".join(l[-3000:] for l in logs)
    return [dict(np.load(o)) for o in outs]


def assemble(ranks):
    """The whole model from the ranks' files + the check that the replicated halves are bit-identical everywhere."""
    for r in ranks[1:]:
        assert np.array_equal(r["Q"], ranks[0]["Q"]), "Q replicas differ after the flush: max %.3e" % np.abs(r["Q"] - ranks[0]["Q"]).max()
        assert np.array_equal(r["Qb"], ranks[0]["Qb"]), "Qb replicas differ after the flush"
    order = sorted(ranks, key=lambda r: int(r["u0"]))
    return np.concatenate([r["P"] for r in order]), ranks[0]["Q"], ranks[0]["Qb"]


GRAD_SPECS = {
    "adagrad": dict(kind="bpr", d=48, epochs=3, lr=0.03, opt=dict(optimizer="adagrad", num_negative_samples=2), modes=dict(chunk=64)),
    "adam_pcn": dict(kind="bpr", d=48, epochs=3, lr=0.03, opt=dict(optimizer="adam", per_coordinate_normalize=True), modes=dict(chunk=64)),
    "warp": dict(kind="warp", d=64, epochs=3, modes=dict(chunk=64)),
}


@pytest.mark.parametrize("name,world", [("adagrad", 2), ("adagrad", 3), ("adam_pcn", 2), ("warp", 2), ("warp", 3), ("adagrad", 8)])
def test_gradient_paths_n_ranks_equal_one_gpu(tmp_path, name, world):
    """algo.cc:382: one optimizer step per epoch from the epoch's summed gradients.  Sharding the users and summing the item-side
    gradient deltas over the ranks before the (then identical) step is the single-GPU computation in another summation order --
    from the FIRST epoch on (the delta base Z must be captured before anything is accumulated)."""
    spec = dict(GRAD_SPECS[name], scenario="sgd", chunks=2)
    ranks = run_world(tmp_path, world, spec)
    assert all(int(r["exchanges"]) == spec["epochs"] for r in ranks)
    P, Q, Qb = assemble(ranks)
    ref = W.run_sgd_rank(spec, 0, 1, None)
    for got, want, f in ((P, ref["P"], "P"), (Q, ref["Q"], "Q"), (Qb, ref["Qb"], "Qb")):
        assert H.relerr(got, want) < 1e-4, (f, H.relerr(got, want))


def restated_protocol(spec, world, weights):
    """exchange_begin / exchange_finish (blocking) in numpy over `world` handles WITHOUT a communicator, one after the other:
        S_r = Q_r - Z;  R = ((S_0 + S_1) + ...) in rank order;  Z <- Z + w R;  Q_r <- Z   (fp32, sgd_base.hip delta_*_kernel)
    w = 1 (plain sum) or the saturation weights of exchange_weight_kernel in float64."""
    csr, opt, P, Q0, Qb0 = W.sgd_problem(spec)
    hs = []
    for r in range(world):
        Qr, Qbr = Q0.copy(), Qb0.copy()
        obj, Pl, (u0, u1, ip) = W.make_sgd(spec, csr, opt, P, Qr, Qbr, r, world, None)
        obj.sync_every_epoch = True
        hs.append((obj, Pl, Qr, Qbr, ip, W.chunk_edges(u1 - u0, spec.get("chunks", 1), ragged=spec.get("ragged_rank") == r), u0))
    Z, Zb = Q0.copy(), Qb0.copy()
    cnt = np.bincount(csr.keys, minlength=csr.num_items).astype(np.float64)
    n_neg = opt["num_negative_samples"]
    glob = float(csr.nnz) * n_neg
    for _ in range(spec["epochs"]):
        for k in range(spec.get("chunks", 1)):
            S, Sb, triples = [], [], []
            for (obj, Pl, Qr, Qbr, ip, edges, u0) in hs:
                Qr[:], Qbr[:] = Z, Zb
                obj.synchronize(False)
                a, b = int(edges[k]), int(edges[k + 1])
                _, n = obj.add_jobs(a, b, ip, None)
                obj.synchronize(True)
                S.append(Qr - Z)
                Sb.append(Qbr - Zb)
                triples.append(np.float32(n))
            R, Rb = S[0].copy(), Sb[0].copy()
            t = np.float32(triples[0])
            for r in range(1, world):
                R += S[r]
                Rb += Sb[r]
                t = np.float32(t + triples[r])
            if weights:
                share = float(t) / world / glob
                lr = float(np.float32(np.float32(opt["lr"]) * world)) / world       # constant lr in these cases (min_lr == lr)
                m = (cnt * n_neg + glob / csr.num_items) * share

                def w_of(x):
                    x = np.asarray(x, dtype=np.float64)
                    out = np.ones_like(x)
                    nz = x > 1e-9
                    out[nz] = -np.expm1(-world * x[nz]) / (world * -np.expm1(-x[nz]))
                    return out.astype(np.float32)
                Wq, Wb = w_of(lr * 0.025 * m), w_of(lr * 0.25 * m)
                Z = Z + Wq[:, None] * R
                Zb = Zb + Wb[:, None] * Rb
            else:
                Z = Z + R
                Zb = Zb + Rb
        for (obj, Pl, Qr, Qbr, ip, edges, u0) in hs:
            obj.update_parameters()
    order = sorted(hs, key=lambda h: h[6])
    return np.concatenate([h[1] for h in order]), Z, Zb


SEQ = dict(scenario="sgd", kind="bpr", d=40, epochs=3, modes=dict(sequential=1))


@pytest.mark.parametrize("world,chunks,ragged", [(2, 1, None), (3, 2, None), (2, 3, 1)])
def test_sgd_plain_sum_is_the_restated_protocol_bit_for_bit(tmp_path, world, chunks, ragged):
    """Deterministic walk, blocking exchange, combination weight 1: every rank's delta is applied exactly once everywhere.
    `ragged`: that rank's first chunk is EMPTY -- it must still enter the call's collective (with a zero delta)."""
    spec = dict(SEQ, chunks=chunks, lr=0.05, min_lr=0.01, modes=dict(sequential=1, comm_stiffness=0, comm_stiffness_q=0))
    if ragged is not None:
        spec["ragged_rank"] = ragged
    ranks = run_world(tmp_path, world, spec)
    assert all(int(r["exchanges"]) == spec["epochs"] * chunks for r in ranks), [int(r["exchanges"]) for r in ranks]
    P, Q, Qb = assemble(ranks)
    rP, rQ, rQb = restated_protocol(spec, world, weights=False)
    assert np.array_equal(Q, rQ) and np.array_equal(Qb, rQb) and np.array_equal(P, rP), (np.abs(Q - rQ).max(), np.abs(P - rP).max())


def test_sgd_saturation_weights_are_the_same_on_every_rank(tmp_path):
    """Default stiffness: the per-row weights come from the all-reduced interval sizes / learning rates, so Z advances by the same
    arithmetic everywhere (bit-identical replicas, checked by `assemble`) and matches the float64 restatement of the formula."""
    spec = dict(SEQ, chunks=2, lr=0.05, min_lr=0.05)
    ranks = run_world(tmp_path, 2, spec)
    P, Q, Qb = assemble(ranks)
    rP, rQ, rQb = restated_protocol(spec, 2, weights=True)
    for got, want in ((P, rP), (Q, rQ), (Qb, rQb)):
        assert H.relerr(got, want) < 1e-5, H.relerr(got, want)
    # ... and it is NOT the plain sum (the weights do something on this problem)
    sP, sQ, sQb = restated_protocol(dict(spec, modes=dict(sequential=1, comm_stiffness=0, comm_stiffness_q=0)), 2, weights=False)
    assert H.relerr(Qb, sQb) > 1e-4


@pytest.mark.parametrize("modes,per_call,world", [({}, 1, 2), (dict(comm_segments=3), 3, 2), (dict(comm_segments=3), 3, 3), (dict(comm_segments=3), 3, 8),
                                                  ({}, 1, 8)])
def test_hogwild_item_major_n_ranks(tmp_path, modes, per_call, world):
    """The throughput walk (item-major, per-XCD replicas) with its exchange blocking (default) and pipelined three deep, on 2, 3
    and 8 ranks: replicas bit-identical after the flush, every call made its exchange points, and the N-rank model ranks as well
    as the one-GPU model on a planted problem (local SGD with summed deltas, statistical parity: SURVEY 8(e))."""
    from buffalo_amd import synth
    spec = dict(scenario="sgd", kind="bpr", d=16, epochs=30, U=600, I=400, density=0.06, data_seed=7, lr=0.05, min_lr=0.01,
                opt=dict(reg_u=0.01, reg_i=0.01, reg_j=0.01, reg_b=0.01), modes=modes)
    ranks = run_world(tmp_path, world, spec, timeout=400)
    assert all(spec["epochs"] * per_call <= int(r["exchanges"]) <= spec["epochs"] * (per_call + 1) for r in ranks), [int(r["exchanges"]) for r in ranks]
    P, Q, Qb = assemble(ranks)
    assert np.isfinite(P).all() and np.isfinite(Q).all() and np.isfinite(Qb).all()
    one = W.run_sgd_rank(dict(spec, modes={}), 0, 1, None)
    csr = W.sgd_problem(spec)[0]
    # how well each model ranks the training positives themselves (no held-out split in this problem): mean AUC-like margin
    def fit(P_, Q_, Qb_):
        d = 16
        s = P_[:, :d] @ Q_[:, :d].T + Qb_[:, 0][None, :]
        rows = csr.rows()
        pos = s[rows, csr.keys].mean()
        return pos - s.mean()
    f2, f1 = fit(P, Q, Qb), fit(one["P"], one["Q"], one["Qb"])
    n2, n1 = np.linalg.norm(Q), np.linalg.norm(one["Q"])
    print("\nHogwild walk, %d ranks, %s: fit %.4f vs one GPU %.4f (ratio %.3f), |Q| ratio %.3f" % (world, modes or "blocking", f2, f1, f2 / f1, n2 / n1))
    assert f1 > 0 and f2 > 0.8 * f1, (f2, f1)
    assert 0.7 < n2 / n1 < 1.4, (n2, n1)


@pytest.mark.parametrize("world,modes", [(3, dict(sequential=1)), (8, dict(sequential=1)), (3, dict(comm_segments=3)), (8, {})])
def test_replicas_stay_bit_identical_under_ring_order_sums(tmp_path, world, modes):
    """RCCL's ring all-reduce does not sum in rank order: each segment of the buffer is accumulated around the ring from a different
    starting rank, and every rank receives the same result.  The exchange protocol needs only the second half of that sentence.
    With the test transport summing in ring order (BFH_COMM_SHM_ORDER=ring): the replicas are still bit-identical on every rank
    (`assemble`), and the model differs from the rank-order run by fp32 summation order only (deterministic walk) / stays the
    same model statistically (Hogwild walk)."""
    spec = dict(scenario="sgd", kind="bpr", d=16, epochs=6, U=600, I=400, density=0.06, data_seed=7, lr=0.05, min_lr=0.01,
                opt=dict(reg_u=0.01, reg_i=0.01, reg_j=0.01, reg_b=0.01), modes=modes)
    ring = run_world(tmp_path / "ring", world, spec, timeout=400, order="ring")
    Pr, Qr, Qbr = assemble(ring)                      # bit-identical replicas under ring-order sums
    rank_order = run_world(tmp_path / "rank", world, spec, timeout=400)
    Po, Qo, Qbo = assemble(rank_order)
    e = max(H.relerr(Qr, Qo), H.relerr(Qbr, Qbo), H.relerr(Pr, Po))
    print("\n%d ranks, %s: ring-order vs rank-order sums, model distance %.2e" % (world, modes, e))
    if modes.get("sequential"):
        assert 0 < e < 1e-4, e                        # not the same bits (the order did change), the same numbers
    else:
        assert e < 0.5, e                             # Hogwild: another legal interleaving of the same updates


@pytest.mark.parametrize("world,optimizer,d", [(2, "manual_cg", 32), (3, "llt", 20), (2, "ialspp", 128)])
def test_als_publish_rows_equals_one_gpu_bit_for_bit(tmp_path, world, optimizer, d):
    """Rows are solved by exactly one rank from identical inputs and travel as raw bytes: P, Q after two epochs are the
    single-GPU factors bit for bit on every rank; the loss sums agree up to the order of N partial sums."""
    spec = dict(scenario="als", epochs=2, optimizer=optimizer, d=d)
    ranks = run_world(tmp_path, world, spec)
    ref = W.run_als_rank(spec, 0, 1, None)
    for r in ranks:
        assert np.array_equal(r["P"], ref["P"]) and np.array_equal(r["Q"], ref["Q"])
        assert np.allclose(r["losses"], ref["losses"], rtol=1e-9, atol=0)
