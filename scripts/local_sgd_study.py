"""Multi-rank Hogwild exchange rules at BASELINE scale, simulated on the CPU oracle (no GPU needed): `world` ranks (one
oracle each, its user shard, its replica of Q / Qb) run their walks one after the other between exchange points -- exactly
the semantics of the blocking / pipelined delta exchange of buffalo_amd/dist.py (= csrc/sgd_base.hip exchange_begin /
exchange_finish), whatever the wall-clock overlap on real GPUs.  Reports sampled loss, norms and top-10 overlap against
the single-process run.   usage: python scripts/local_sgd_study.py [lr] [min_lr] [epochs] [world]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from buffalo_amd import synth
from buffalo_amd.dist import shard_csr
from oracle import oracle as orc
orc.build()
lr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.002
min_lr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0001
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
world = int(sys.argv[4]) if len(sys.argv) > 4 else 8
csr = bench.load_matrix("ml20m", 7)
U, I, nnz = csr.num_users, csr.num_items, csr.nnz
rng = np.random.default_rng(0)
eu = rng.integers(0, U, 20000).astype(np.int32)
beg = np.where(eu == 0, 0, csr.indptr[np.maximum(eu, 1) - 1])
ep = np.ascontiguousarray(csr.keys[beg].astype(np.int32))
en = rng.integers(0, I, 20000).astype(np.int32)
users = np.random.default_rng(1).choice(U, 2000, replace=False)


def top10(P, Q, Qb):
    return np.argsort(-(P[users] @ Q.T + Qb.reshape(1, -1)), axis=1)[:, :10]


K0 = float(os.environ.get("K0", "0.25"))          # stiffness of the bias dynamics (curvature of the logistic loss)
KQ = float(os.environ.get("KQ", os.environ.get("K0", "0.25")))   # stiffness assumed for the factor rows
item_cnt = np.bincount(csr.keys, minlength=I).astype(np.float64)


def weights(world, exchanges, lr_now, k0=None):
    """Saturation-aware combination (csrc/sgd_base.hip exchange weights): a row that receives m updates per rank and
    interval contracts by exp(-x), x = lr * k0 * m, towards its local equilibrium; N such deltas from the same start
    combine like ONE run of N m updates when scaled by (1 - exp(-N x)) / (N (1 - exp(-x))): 1 (sum) for cold rows,
    1/N (mean) for saturated ones."""
    m = (item_cnt + nnz / I) / (world * exchanges)          # positive + expected negative updates per rank and interval
    x = np.maximum(lr_now * (K0 if k0 is None else k0) * m, 1e-12)
    return ((1.0 - np.exp(-world * x)) / (world * (1.0 - np.exp(-x)))).astype(np.float32)[:, None]


def run(world, exchanges, pipelined, sat=False):
    opt = bench.bpr_options(epochs, lr=lr, min_lr=min_lr, accelerator=False, num_workers=1)
    P, Q, Qb = synth.init_factors(U, I, 128, seed=7)
    ranks = []
    for r in range(world):
        u0, u1, ip, keys, off = shard_csr(csr.indptr, csr.keys, r, world)
        Pl, Ql, Qbl = np.ascontiguousarray(P[u0:u1]), Q.copy(), Qb.copy()
        o = orc.OracleBPRMF()
        assert o.init(bench.write_opt(opt))
        o.initialize_model(Pl, Ql, Qbl, nnz)
        o.set_cumulative_table(np.zeros(I, np.int64), I)
        o.set_modes(sampler="counter", pos_order="csr", inline=True)
        o.set_shard(off, world)
        o.launch_workers()
        ranks.append((o, Pl, Ql, Qbl, u0, u1, ip, keys))
    Z, Zb, pend = Q.copy(), Qb.copy(), None
    for e in range(epochs):
        for x in range(exchanges):
            for (o, Pl, Ql, Qbl, u0, u1, ip, keys) in ranks:
                n = u1 - u0
                a, b = n * x // exchanges, n * (x + 1) // exchanges
                kb, ke = (0 if a == 0 else int(ip[a - 1])), (int(ip[b - 1]) if b > 0 else 0)
                if b > a:
                    o.add_jobs(a, b, ip, np.ascontiguousarray(keys[kb:ke]))
            if world == 1:
                continue
            if pend is not None:                                   # exchange_finish(progressed): the other ranks' part lands
                Sp, Sbp, R, Rb = pend
                Z += R
                Zb += Rb
                for k, rk in enumerate(ranks):
                    rk[2][:] += R - Sp[k]
                    rk[3][:] += Rb - Sbp[k]
                pend = None
            S, Sb = [rk[2] - Z for rk in ranks], [rk[3] - Zb for rk in ranks]
            R, Rb = sum(S), sum(Sb)
            if sat:
                frac = (e + (x + 0.5) / exchanges) / epochs
                lr_now = max(min_lr, lr - (lr - min_lr) * frac)
                R, Rb = R * weights(world, exchanges, lr_now, KQ), Rb * weights(world, exchanges, lr_now, K0)
            if pipelined:
                pend = (S, Sb, R, Rb)
            else:
                Z += R
                Zb += Rb
                for rk in ranks:
                    rk[2][:] = Z
                    rk[3][:] = Zb
        for rk in ranks:
            rk[0].update_parameters()
    if pend is not None:
        Z += pend[2]
        Zb += pend[3]
    if world == 1:
        Z, Zb = ranks[0][2], ranks[0][3]
    Pall = np.concatenate([rk[1] for rk in ranks])
    o = orc.OracleBPRMF()
    assert o.init(bench.write_opt(opt))
    o.initialize_model(Pall, Z, Zb, nnz)
    return {"loss": o.compute_loss(eu, ep, en), "P": float(np.linalg.norm(Pall)), "Q": float(np.linalg.norm(Z)), "Qb": float(np.linalg.norm(Zb))}, top10(Pall, Z, Zb)


out = {"lr": lr, "min_lr": min_lr, "epochs": epochs, "world": world, "runs": {}}
t0 = time.time()
base, tb = run(1, 1, False)
out["runs"]["single_process"] = base
print("single", base, "%.0f s" % (time.time() - t0), flush=True)
CONFIGS = [(1, False, False), (1, True, False), (4, True, False), (1, False, True), (1, True, True), (4, True, True)]
if os.environ.get("ONLY_SAT"):
    CONFIGS = [c for c in CONFIGS if c[2]]
if os.environ.get("CONFIGS"):      # e.g. CONFIGS="2,1,1;8,1,1" = (exchanges, pipelined, saturation weights)
    CONFIGS = [tuple(int(v) for v in c.split(",")) for c in os.environ["CONFIGS"].split(";")]
    CONFIGS = [(a, bool(b), bool(c)) for a, b, c in CONFIGS]
for exchanges, pipelined, sat in CONFIGS:
    m, t = run(world, exchanges, pipelined, sat)
    m["top10_overlap_vs_single"] = float(np.mean([len(set(a) & set(b)) / 10 for a, b in zip(t, tb)]))
    name = "world%d_%dx_%s%s" % (world, exchanges, "pipelined" if pipelined else "blocking", "_saturation_weights_kb%g_kq%g" % (K0, KQ) if sat else "")
    out["runs"][name] = m
    print(name, m, "%.0f s" % (time.time() - t0), flush=True)
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_local_sgd_study_lr%g_world%d%s.json" % (lr, world, os.environ.get("TAG", ""))), "w"), indent=1)
