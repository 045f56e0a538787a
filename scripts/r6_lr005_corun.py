"""Round-6 study (b) of the BPR lr cliff: sum or max?  At lr 0.05 every negative of the item-major walk is a coherent row read + an fp32 row
atomic on the chip-wide matrix, and a launch costs the walk's 4.0-4.3 ms PLUS the 4.3 ms the same atomics take on their own.
Here the two halves run (1) alone and (2) side by side on two streams -- the walk with its negatives' atomics masked (`im_study` = 1: timing only,
wrong results), the atomics as scripts/micro/atomics_corun.hip on its own matrix.  If the pair finishes in about the time of the slower half, the
atomic units and the walk's load path are separate resources and the additive cost is serialisation inside a wave; if it takes the sum, it is the chip.
Also: the unmasked walk at 16 / 20 / 24 waves per CU (latency-bound work speeds up with residency, throughput-bound work does not).

env: EPOCHS (5).  Needs scripts/micro/libatomics_corun.so (the header of the .hip says how to build it)."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import test_bpr_gate_gpu as G  # noqa: E402
from buffalo_amd import synth  # noqa: E402
from buffalo_amd.backend import CyBPR  # noqa: E402

from buffalo_amd import _lib  # noqa: E402
_lib.lib()   # first: it loads the HIP runtime this process shares with torch; the micro library's libamdhip64 then resolves to the same objects
lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "micro", "libatomics_corun.so"))
lib.corun_start.argtypes = [ctypes.c_longlong, ctypes.c_int, ctypes.c_int]
lib.corun_wait.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int]
EPOCHS = int(os.environ.get("EPOCHS", "5"))
csr = G._csr()
U, I, nnz = csr.num_users, csr.num_items, csr.nnz
assert lib.corun_init(I) == 0
ROWS = nnz // 2      # the negatives of one launch of the walk (two launches per epoch)
out = []


def corun_alone(wpc, launches=6):
    assert lib.corun_start(ROWS, launches, wpc) == 0
    ms = (ctypes.c_float * launches)()
    assert lib.corun_wait(ms, launches) == 0
    return [round(float(x), 3) for x in ms]


def walk(modes, corun_wpc=0):
    opt = bench.bpr_options(EPOCHS + 2, lr=0.05, min_lr=0.05)
    P, Q, Qb = synth.init_factors(U, I, G.D, seed=7)
    obj = CyBPR()
    assert obj.init(bench.write_opt(dict(opt, accelerator=True)))
    obj.sync_every_epoch = False
    for k, v in modes.items():
        obj.set_mode(k, v)
    obj.initialize_model(P, Q, Qb, nnz, True)
    obj.set_cumulative_table(np.zeros(I, np.int64), I)
    obj.set_resident_csr(csr.indptr, csr.keys)
    for _ in range(2):   # warm-up: the regrouping, the flags, the pre-drawn negatives
        obj.add_jobs(0, U, csr.indptr, None)
        obj.update_parameters()
    s0 = obj.stats()
    launches = 0
    if corun_wpc:
        launches = 4 * EPOCHS + 8    # enough launches to outlast the walk's epochs
        assert lib.corun_start(ROWS, launches, corun_wpc) == 0
    for _ in range(EPOCHS):
        obj.add_jobs(0, U, csr.indptr, None)
        obj.update_parameters()
    s1 = obj.stats()
    co = None
    if corun_wpc:
        ms = (ctypes.c_float * launches)()
        assert lib.corun_wait(ms, launches) == 0
        co = [round(float(x), 3) for x in ms]
    k = (s1["kernel_ms"] - s0["kernel_ms"]) / max(1, s1["launches"] - s0["launches"])
    del obj
    return round(k, 3), co


for wpc in (8, 12, 16):
    r = {"what": "atomics alone", "waves_per_cu": wpc, "rows_per_launch": ROWS, "ms_per_launch": corun_alone(wpc)}
    out.append(r); print(json.dumps(r), flush=True)
for modes in ({}, {"waves_per_cu": 16}, {"waves_per_cu": 20}, {"waves_per_cu": 24}, {"im_study": 1}, {"im_study": 1, "waves_per_cu": 16}):
    k, _ = walk(modes)
    r = {"what": "walk alone", "modes": modes, "kernel_ms_per_launch": k}
    out.append(r); print(json.dumps(r), flush=True)
for wpc_walk in (16, 20):
    for wpc_co in (8, 12):
        k, co = walk({"im_study": 1, "waves_per_cu": wpc_walk}, corun_wpc=wpc_co)
        r = {"what": "masked walk + atomics side by side", "walk_waves_per_cu": wpc_walk, "atomics_waves_per_cu": wpc_co, "walk_kernel_ms_per_launch": k,
             "atomics_ms_per_launch": co}
        out.append(r); print(json.dumps(r), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r6_lr005_corun.json"), "w"), indent=1)
