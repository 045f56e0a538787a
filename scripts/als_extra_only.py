import json, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
csr = bench.load_matrix("ml20m", 7)
out = bench.extra_als(csr, 7, epochs=3, cpu=False)
print(json.dumps(out["mfma"]))
print(out["epoch_ms"], out["kernel_ms_per_epoch"])
