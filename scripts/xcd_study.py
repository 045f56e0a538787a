"""Policy 2 (per-XCD item-factor replicas, hot rows on atomics) against policy 1 (atomics everywhere):
    timing  -- knob sweep on the bench workload (ML-20M-shaped, d=128, bench.py's options);
    planted -- ranking quality (NDCG@10) on a popularity-skewed planted matrix vs the CPU oracle;
    ml20m   -- sampled / fixed-triple loss and factor norms at BASELINE scale, lr=0.05.
Writes gpurun_out/xcd_study.json.  Usage: python scripts/xcd_study.py [timing] [planted] [ml20m]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import bpr_options, load_matrix, write_opt  # noqa: E402
from buffalo_amd import synth  # noqa: E402
from buffalo_amd.backend import CyBPR  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out", "xcd_study.json")
out = {}
if os.path.exists(OUT):
    out = json.load(open(OUT))


def save():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    json.dump(out, open(OUT, "w"), indent=1)


def make(opt, modes, P, Q, Qb, csr):
    obj = CyBPR()
    path = write_opt(opt)
    assert obj.init(path)
    os.unlink(path)
    obj.sync_every_epoch = False
    for k, v in modes.items():
        obj.set_mode(k, v)
    obj.initialize_model(P, Q, Qb, csr.nnz, True)
    obj.set_cumulative_table(np.zeros(csr.num_items, np.int64), csr.num_items)
    obj.set_resident_csr(csr.indptr, csr.keys)
    return obj


def timing():
    csr = load_matrix("ml20m", 7)
    U, I, nnz = csr.num_users, csr.num_items, csr.nnz
    res = out.setdefault("timing", {})
    cfgs = [dict(), dict(im_presample=0), dict(im_blocks=1), dict(im_blocks=4), dict(xcd_sync_updates=1 << 25)]
    steps, warm = 6, 2
    for modes in cfgs:
        name = ",".join("%s=%s" % kv for kv in modes.items()) or "default(hogwild_atomic=3)"
        P, Q, Qb = synth.init_factors(U, I, 128, seed=7)
        obj = make(bpr_options(steps + warm), modes, P, Q, Qb, csr)
        for _ in range(warm):
            obj.add_jobs(0, U, csr.indptr, None)
            obj.update_parameters()
        obj.reset_stats()
        t0 = time.perf_counter()
        for _ in range(steps):
            obj.add_jobs(0, U, csr.indptr, None)
            obj.update_parameters()
        dt = (time.perf_counter() - t0) / steps
        st = obj.stats()
        res[name] = {"epoch_ms": dt * 1e3, "update_kernel_ms_per_epoch": st["kernel_ms"] / steps, "aux_ms_per_epoch": st["aux_ms"] / steps,
                     "launches_per_epoch": st["launches"] / steps, "merges_per_epoch": st["merges"] / steps,
                     "frac_of_8TBs": (24 * 128 + 20) * nnz / dt / 8e12}
        print("timing", name, json.dumps(res[name]), flush=True)
        del obj
        save()


def planted():
    import helpers as H
    from oracle import oracle as orc
    orc.build()
    U, I, d, epochs = 12000, 4000, 32, 10
    csr, vali = synth.planted(U, I, d_true=8, density=0.03, seed=7, popularity=1.5)
    cnt = np.bincount(csr.keys, minlength=I)
    res = out.setdefault("planted", {"shape": [U, I, int(csr.nnz)], "d": d, "epochs": epochs,
                                     "item_count_max_mean": [int(cnt.max()), float(cnt.mean())]})
    opt = bpr_options(epochs, d=d, lr=0.05, min_lr=0.01, reg_u=0.01, reg_i=0.01, reg_j=0.01, reg_b=0.01, num_workers=64)
    P0, Q0, Qb0 = synth.init_factors(U, I, d, seed=7)
    res["untrained"] = H.ndcg_at_k(P0, Q0, csr, vali, Qb=Qb0)
    t0 = time.perf_counter()
    Po, Qo, Qbo = P0.copy(), Q0.copy(), Qb0.copy()
    H.run_oracle_sgd(orc.OracleBPRMF, dict(opt, accelerator=False), csr, Po, Qo, Qbo, epochs=epochs)
    res["cpu_oracle_64_threads"] = {"ndcg": H.ndcg_at_k(Po, Qo, csr, vali, Qb=Qbo), "seconds": time.perf_counter() - t0,
                                    "norms": [float(np.linalg.norm(Po)), float(np.linalg.norm(Qo))]}
    print("planted cpu", res["cpu_oracle_64_threads"], flush=True)
    save()
    for modes in (dict(hogwild_atomic=1), dict(hogwild_atomic=3), dict(hogwild_atomic=3, xcd_sync_updates=1 << 23)):
        name = ",".join("%s=%s" % kv for kv in modes.items())
        if modes.get("sequential") and csr.nnz > 3_000_000:
            continue
        P, Q, Qb = H.pad(P0, 32), H.pad(Q0, 32), Qb0.copy()
        t0 = time.perf_counter()
        H.run_hip_sgd(CyBPR, opt, csr, P, Q, Qb, epochs=epochs, modes=modes, resident=True)
        res[name] = {"ndcg": H.ndcg_at_k(P[:, :d], Q[:, :d], csr, vali, Qb=Qb), "seconds": time.perf_counter() - t0,
                     "norms": [float(np.linalg.norm(P)), float(np.linalg.norm(Q))]}
        print("planted", name, res[name], flush=True)
        save()


def ml20m():
    csr = load_matrix("ml20m", 7)
    U, I, nnz = csr.num_users, csr.num_items, csr.nnz
    epochs = 4
    rng = np.random.default_rng(0)
    eu = rng.integers(0, U, 4000).astype(np.int32)
    ep = np.array([csr.row(int(u))[0][0] for u in eu], dtype=np.int32)
    en = rng.integers(0, I, 4000).astype(np.int32)
    cnt = np.bincount(csr.keys, minlength=I)
    head = np.argsort(-cnt)[:200]
    for lr in (0.05, 0.002):
        res = out.setdefault("ml20m_lr%g" % lr, {"epochs": epochs, "lr": lr})
        for modes in (dict(hogwild_atomic=1), dict(), dict(im_blocks=4), dict(im_blocks=8), dict(im_blocks=16), dict(prefetch=1)):
            name = ",".join("%s=%s" % kv for kv in modes.items()) or "default(hogwild_atomic=3)"
            P, Q, Qb = synth.init_factors(U, I, 128, seed=7)
            obj = make(bpr_options(epochs, lr=lr, min_lr=lr, compute_loss_on_training=True), modes, P, Q, Qb, csr)
            tr, ev = [], [obj.compute_loss(eu, ep, en)]
            for _ in range(epochs):
                loss, n = obj.add_jobs(0, U, csr.indptr, None)
                obj.update_parameters()
                tr.append(loss / n)
                ev.append(obj.compute_loss(eu, ep, en))
            obj.synchronize(True)
            res[name] = {"train_loss": tr, "eval_loss": ev, "P_norm": float(np.linalg.norm(P)), "Q_norm": float(np.linalg.norm(Q)),
                         "Q_head200_norm": float(np.linalg.norm(Q[head])), "Qb_norm": float(np.linalg.norm(Qb))}
            print("ml20m lr=%g" % lr, name, json.dumps(res[name]), flush=True)
            del obj
            save()
        if os.environ.get("WITH_CPU", "0") == "1" and lr == 0.002:
            from oracle import oracle as orc
            orc.build()
            P, Q, Qb = synth.init_factors(U, I, 128, seed=7)
            o = orc.OracleBPRMF()
            assert o.init(write_opt(bpr_options(epochs, lr=lr, min_lr=lr, accelerator=False, num_workers=64)))
            o.initialize_model(P, Q, Qb, nnz)
            o.set_cumulative_table(np.zeros(I, np.int64), I)
            o.launch_workers()
            ev = [o.compute_loss(eu, ep, en)]
            for e in range(epochs):
                o.add_jobs(0, U, csr.indptr, csr.keys)
                prev = -1
                while True:
                    o.wait_until_done()
                    cur = o.stats()["samples"]
                    if cur == prev and cur >= (e + 1) * nnz:
                        break
                    prev = cur
                    time.sleep(0.05)
                o.update_parameters()
                ev.append(o.compute_loss(eu, ep, en))
            o.join()
            res["cpu_oracle_64_threads"] = {"eval_loss": ev, "P_norm": float(np.linalg.norm(P)), "Q_norm": float(np.linalg.norm(Q)),
                                            "Q_head200_norm": float(np.linalg.norm(Q[head])), "Qb_norm": float(np.linalg.norm(Qb))}
            print("ml20m lr=%g cpu" % lr, json.dumps(res["cpu_oracle_64_threads"]), flush=True)
            save()


if __name__ == "__main__":
    todo = sys.argv[1:] or ["timing", "planted", "ml20m"]
    for t in todo:
        {"timing": timing, "planted": planted, "ml20m": ml20m}[t]()
    save()
