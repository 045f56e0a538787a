"""Randomised differential run: the oracle against the reference's own SGD sources on the stand-ins (TEST INFRASTRUCTURE, not in the suite).

    BUFFALO_REF_SGD_EXACT=1 BUFFALO_ORACLE_LIB=oracle/_ref/libbuffalo_oracle_exact.so python tests/golden/fuzz_reference_sources.py SEED COUNT

Random shapes, widths (1 .. 100), epochs, chunk counts and option sets for BPRMF (optimizer, bias, negatives, sampling power, verify_neg,
update_i / update_j, per-coordinate normalisation, regularisers, seed, rate) and WARP (optimizer, trial limit, threshold, score function,
normalisation, regularisers); one worker, constant rate.  Built without FP contraction the two must agree to the bit: prints every
mismatch.  Results of the runs made in round 2 are in profiles/r02_reference_code_run_in_container_cpu.txt.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import oracle, ref_sgd
import helpers as H
from conftest import bpr_opt, warp_opt, tiny_csr
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
def run(cls, opt, csr, P, Q, Qb, epochs, chunks):
    o = cls(); path = H.write_opt(opt); assert o.init(path); os.unlink(path)
    o.initialize_model(P, Q, Qb, csr.nnz)
    o.set_cumulative_table(H.cum_table(csr, opt), csr.num_items)
    o.launch_workers()
    for _ in range(epochs):
        for a, b in H.chunks_of(csr, chunks):
            keys, _ = H.chunk_arrays(csr, a, b); o.add_jobs(a, b, csr.indptr, keys)
        o.wait_until_done(); time.sleep(0.05); o.update_parameters()
    o.join()
    users=np.arange(8,dtype=np.int32); pos=np.arange(8,dtype=np.int32); neg=np.arange(8,16,dtype=np.int32)
    return o.compute_loss(users,pos,neg)
bad=0; n=int(sys.argv[2]) if len(sys.argv)>2 else 40
for t in range(n):
    U,I=int(rng.integers(17,90)),int(rng.integers(16,120)); csr=tiny_csr(U=U,I=I,density=float(rng.uniform(0.05,0.4)),seed=int(rng.integers(1000)))
    d=int(rng.choice([1,3,8,20,33,64,100])); epochs=int(rng.integers(1,4)); chunks=int(rng.integers(1,4))
    if rng.random()<0.6:
        kw=dict(optimizer=str(rng.choice(["sgd","adagrad","adam"])), use_bias=bool(rng.random()<0.7), num_negative_samples=int(rng.integers(1,4)),
                sampling_power=float(rng.choice([0.0,0.0,0.5,1.0,3.0])), verify_neg=bool(rng.random()<0.7), update_i=bool(rng.random()<0.85), update_j=bool(rng.random()<0.85),
                per_coordinate_normalize=bool(rng.random()<0.3), reg_u=float(rng.choice([0,0.025,0.3])), reg_i=float(rng.choice([0,0.025,0.3])), reg_j=float(rng.choice([0,0.025])), reg_b=float(rng.choice([0,0.025,1.0])),
                random_seed=int(rng.integers(0,1000)))
        lr=float(rng.choice([0.002,0.05,0.3])); opt=bpr_opt(d=d, lr=lr, min_lr=lr, num_iters=epochs, num_workers=1, **kw); ocls,rcls=oracle.OracleBPRMF,ref_sgd.RefBPRMF; name="bpr"
    else:
        kw=dict(optimizer=str(rng.choice(["adagrad","adam"])), max_trials=int(rng.choice([1,2,5,20,500])), threshold=float(rng.choice([0.1,0.5,1.0,50.0])), score_func=str(rng.choice(["dot","l2"])),
                per_coordinate_normalize=bool(rng.random()<0.3), reg_u=float(rng.choice([0,0.01])), reg_i=float(rng.choice([0,0.02])), reg_j=float(rng.choice([0,0.03])), random_seed=int(rng.integers(0,1000)))
        lr=float(rng.choice([0.01,0.05,0.5])); opt=warp_opt(d=d, lr=lr, min_lr=lr, num_iters=epochs, num_workers=1, **kw); ocls,rcls=oracle.OracleWARP,ref_sgd.RefWARP; name="warp"
    g=np.random.default_rng(t)
    sc=float(rng.choice([0.05,0.3,1.0]))
    P0=g.normal(scale=sc,size=(U,d)).astype(np.float32); Q0=g.normal(scale=sc,size=(I,d)).astype(np.float32); Qb0=g.normal(scale=0.1,size=(I,1)).astype(np.float32)
    if name=="warp" or not opt.get("use_bias",False): Qb0*=0
    A=[x.copy() for x in (P0,Q0,Qb0)]; B=[x.copy() for x in (P0,Q0,Qb0)]
    la=run(ocls,opt,csr,*A,epochs,chunks); lb=run(rcls,opt,csr,*B,epochs,chunks)
    ok=all(np.array_equal(a,b) for a,b in zip(A,B)) and (la==lb or (np.isnan(la) and np.isnan(lb)))
    fin=all(np.isfinite(a).all() for a in A)
    if not ok:
        bad+=1; print("MISMATCH", name, U,I,d,epochs,chunks, kw, [float(np.nanmax(np.abs(a-b))) for a,b in zip(A,B)], la, lb, flush=True)
    elif t%10==0: print("ok",t,name,d,kw.get("optimizer"),"finite" if fin else "non-finite", flush=True)
print("fuzz done:", n, "configurations,", bad, "mismatches")
