"""How much does the headline kernel's time move from handle to handle inside ONE process (same data, same code)?  Four handles one after the other,
40 epochs each; prints kernel ms per launch, the device pointers of P / Q (placement) and the shader clock state is left to the box."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from buffalo_amd import synth
from buffalo_amd.backend import CyBPR
csr = bench.load_matrix("ml20m", 7)
U, I, nnz = csr.num_users, csr.num_items, csr.nnz
ITEM_DIV = int(os.environ.get("ITEM_DIV", "1"))     # study: the same interactions on a catalogue ITEM_DIV times smaller (item ids merged in pairs / fours): the per-XCD
if ITEM_DIV > 1:                                    # replicas of the item table shrink with it -- does the per-handle draw depend on the size of the kernel's working set?
    csr = synth.CSR(U, (I + ITEM_DIV - 1) // ITEM_DIV, csr.indptr, (csr.keys // ITEM_DIV).astype(np.int32), csr.vals)
    I = csr.num_items
keep = []
if os.environ.get("PREWARM", "0") != "0":   # allocate and free a large block first: does the FIRST handle then behave like the later ones?
    import torch
    gb = float(os.environ["PREWARM"])
    t = torch.empty(int(gb * (1 << 30)), dtype=torch.uint8, device="cuda")
    t.zero_()
    torch.cuda.synchronize()
    del t
    torch.cuda.empty_cache()
if os.environ.get("WARM_TINY", "0") != "0":    # study: a small throw-away handle first -- device code loaded, sort storage sized, streams created before the measured handle allocates
    rng = np.random.default_rng(3)
    tu, ti, per = 12000, 3000, 40
    tkeys = np.sort(rng.integers(0, ti, size=(tu, per)), axis=1).astype(np.int32).reshape(-1)
    tcsr = synth.CSR(tu, ti, np.arange(1, tu + 1, dtype=np.int64) * per, tkeys, np.ones(tu * per, np.float32))
    tP, tQ, tQb = synth.init_factors(tu, ti, bench.D, seed=3)
    t = CyBPR()
    assert t.init(bench.write_opt(bench.bpr_options(4)))
    t.sync_every_epoch = False
    t.initialize_model(tP, tQ, tQb, tcsr.nnz, True)
    t.set_cumulative_table(np.zeros(ti, np.int64), ti)
    t.set_resident_csr(tcsr.indptr, tcsr.keys)
    for _ in range(3):
        t.add_jobs(0, tu, tcsr.indptr, None); t.update_parameters()
    print("tiny handle: %s" % {k: v for k, v in t.stats().items() if k in ("launches", "samples")}, flush=True)
    if os.environ["WARM_TINY"] == "1":
        del t            # "2": keep it alive
_dummy = []
if int(os.environ.get("DUMMY_STREAMS", "0")) > 0:    # study: HIP streams created (and kept) before the first handle shift which hardware queue the handle's streams land on
    import torch
    _dummy = [torch.cuda.Stream() for _ in range(int(os.environ["DUMMY_STREAMS"]))]
    for st in _dummy:
        with torch.cuda.stream(st):
            torch.zeros(8, device="cuda").add_(1)
    torch.cuda.synchronize()
for rep in range(int(os.environ.get("REPS", "4"))):
    P, Q, Qb = synth.init_factors(U, I, bench.D, seed=7)
    g = CyBPR()
    assert g.init(bench.write_opt(bench.bpr_options(45)))
    g.sync_every_epoch = False
    for k, v in __import__("json").loads(os.environ.get("MODES", "{}")).items():
        g.set_mode(k, v)
    g.initialize_model(P, Q, Qb, nnz, True)
    g.set_cumulative_table(np.zeros(I, np.int64), I)
    g.set_resident_csr(csr.indptr, csr.keys)
    for _ in range(5):
        g.add_jobs(0, U, csr.indptr, None); g.update_parameters()
    g.reset_stats()
    for _ in range(40):
        g.add_jobs(0, U, csr.indptr, None); g.update_parameters()
    st = g.stats()
    ptrs = {n: hex(g.device_buffer(n)[0]) for n in ("P", "Q")}
    print("handle %d: kernel %.3f ms per launch, aux %.3f ms per epoch, %s" % (rep, st["kernel_ms"] / st["launches"], st["aux_ms"] / 40, ptrs), flush=True)
    if os.environ.get("KEEP", "0") == "1":
        keep.append(g)       # keep the allocations: the next handle lands elsewhere
    else:
        del g
