"""als_gramian_kernel (CALS::precompute, als.cc:86-93) on the ML-20M shape, d = 128: device time of FF = F^T F for both sides under "als_gram_waves" (waves
per CU) x "als_gram_upg" (row pairs per trip)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from buffalo_amd import synth
from buffalo_amd.backend import CyALS
U, I, _ = synth.SHAPES["ml20m"]
D = int(sys.argv[1]) if len(sys.argv) > 1 else bench.D
for waves in (4, 8, 12):
    for upg in (4, 8):
        P, Q, _ = synth.init_factors(U, I, D, seed=7)
        g = CyALS()
        assert g.init(bench.write_opt(dict(bench.ALS_OPT, d=D)))
        g.set_mode("als_gram_waves", waves); g.set_mode("als_gram_upg", upg)
        g.initialize_model(P, Q)
        out = []
        for axis in (0, 1):
            g.precompute(axis)
            g.reset_stats()
            for _ in range(5):
                g.precompute(axis)
            out.append(g.stats()["aux_ms"] / 5)
        ff = g.device_tensor("FF", (D, D)).cpu().numpy().astype(np.float64)
        ref = P.astype(np.float64).T @ P.astype(np.float64)
        print("d=%d waves/CU %2d  pairs/trip %d:  FF of the items (axis 0) %.3f ms   of the users (axis 1) %.3f ms   max rel err vs float64 %.2e"
              % (D, waves, upg, out[0], out[1], np.abs(ff - ref).max() / np.abs(ref).max()), flush=True)
        del g
