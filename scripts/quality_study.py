"""Statistical parity of the Hogwild policies at BASELINE scale (ML-20M-shaped, d=128): sampled training loss per epoch +
fixed-sample BPR loss for hogwild_atomic = 3 (item-major default), 1, 2, 0 and the CPU oracle.  (scripts/xcd_study.py is the
newer, knob-sweeping version; this one also runs the 64-thread oracle at lr 0.05.)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import bpr_options, load_matrix, write_opt
from buffalo_amd import synth
from buffalo_amd.backend import CyBPR

EPOCHS = int(os.environ.get("EPOCHS", "4"))
LR = float(os.environ.get("LR", "0.05"))
csr = load_matrix("ml20m", 7)
U, I, nnz = csr.num_users, csr.num_items, csr.nnz
rng = np.random.default_rng(0)
# fixed evaluation triples: (user, one of its positives, a non-positive)
eu = rng.integers(0, U, 4000).astype(np.int32)
ep = np.array([csr.row(int(u))[0][0] for u in eu], dtype=np.int32)
en = rng.integers(0, I, 4000).astype(np.int32)
out = {"epochs": EPOCHS, "lr": LR, "runs": {}}


def opts(**kw):
    return bpr_options(EPOCHS, lr=LR, min_lr=LR, compute_loss_on_training=True, **kw)


for mode in (3, 1, 2, 0):
    P, Q, Qb = synth.init_factors(U, I, 128, seed=7)
    obj = CyBPR()
    assert obj.init(write_opt(opts()))
    obj.sync_every_epoch = False
    obj.set_mode("hogwild_atomic", mode)
    obj.initialize_model(P, Q, Qb, nnz, True)
    obj.set_cumulative_table(np.zeros(I, np.int64), I)
    obj.set_resident_csr(csr.indptr, csr.keys)
    tr, ev, ms = [], [obj.compute_loss(eu, ep, en)], []
    for e in range(EPOCHS):
        t0 = time.perf_counter()
        loss, n = obj.add_jobs(0, U, csr.indptr, None)
        obj.update_parameters()
        ms.append((time.perf_counter() - t0) * 1e3)
        tr.append(loss / n)
        ev.append(obj.compute_loss(eu, ep, en))
    obj.synchronize(True)
    out["runs"]["hip_policy_%d" % mode] = {"train_loss": tr, "eval_loss": ev, "epoch_ms": ms,
                                           "Q_norm": float(np.linalg.norm(Q)), "P_norm": float(np.linalg.norm(P))}
    print(mode, tr, ev, ms, flush=True)

if os.environ.get("WITH_CPU", "1") == "1":
    from oracle import oracle as orc
    P, Q, Qb = synth.init_factors(U, I, 128, seed=7)
    o = orc.OracleBPRMF()
    assert o.init(write_opt(opts(accelerator=False, num_workers=64)))
    o.initialize_model(P, Q, Qb, nnz)
    o.set_cumulative_table(np.zeros(I, np.int64), I)
    o.launch_workers()
    ev = [o.compute_loss(eu, ep, en)]
    for e in range(EPOCHS):
        o.add_jobs(0, U, csr.indptr, csr.keys)
        prev = -1
        while True:                      # drain: wait_until_done only waits for an empty queue
            o.wait_until_done()
            cur = o.stats()["samples"]
            if cur == prev and cur >= (e + 1) * nnz:
                break
            prev = cur
            time.sleep(0.05)
        o.update_parameters()
        ev.append(o.compute_loss(eu, ep, en))
    o.join()
    out["runs"]["cpu_oracle_64_threads"] = {"eval_loss": ev, "Q_norm": float(np.linalg.norm(Q)), "P_norm": float(np.linalg.norm(P))}
    print("cpu", ev, flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "quality_study.json"), "w"), indent=1)
