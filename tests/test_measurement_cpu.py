"""Host-side checks of the measurement plumbing (no GPU): the per-epoch attribution of rocprofv3 launches (scripts/pmc_kernels.py -- a
kernel whose launch count is not a multiple of the epochs once had ALL its launches charged to the last epoch, which turned a 21 ms
sort into "11 % of the epoch") and the three byte figures of bench.warp_epoch_row."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _trace(path, launches):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        t = 0
        for name, ns in launches:
            w.writerow([name, t, t + ns])
            t += ns + 10


def _counters(path, launches, counter):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        for i, (name, v) in enumerate(launches):
            w.writerow([i, name, counter, v])


def test_last_epoch_attribution(tmp_path):
    epochs = 3
    # per epoch: 1 walk launch (slower in the last epoch), 2 gathers; the sort runs 5 launches per epoch + 5 once for a static list;
    # a fill kernel runs once per model
    launches, fetch = [], []
    launches += [("void ns::sort_kernel<1>(int)", 100)] * 5 + [("fill(int)", 999)]
    for e in range(epochs):
        launches += [("void ns::walk<4, 4>(P)", 1_000_000 * (e + 1))]
        launches += [("void ns::gather<8>(G)", 2_000_000)] * 2
        launches += [("void ns::sort_kernel<1>(int)", 100_000)] * 5
    fetch = [(n, 10.0 if "walk" in n else 1.0) for n, _ in launches]
    root = str(tmp_path / "prof")
    _trace(os.path.join(root, "stats", "x", "p_kernel_trace.csv"), launches)
    _counters(os.path.join(root, "pmc_fetch", "x", "p_counter_collection.csv"), fetch, "FETCH_SIZE")
    _counters(os.path.join(root, "pmc_write", "x", "p_counter_collection.csv"), [(n, 2.0) for n, _ in launches], "WRITE_SIZE")
    out = str(tmp_path / "out.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_kernels.py"), root, str(epochs), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = json.load(open(out))
    k = d["kernels"]
    assert k["ns::walk<4, 4>"]["launches_per_epoch"] == 1 and abs(k["ns::walk<4, 4>"]["ms"] - 3.0) < 1e-9          # the LAST epoch's launch
    assert k["ns::gather<8>"]["launches_per_epoch"] == 2 and abs(k["ns::gather<8>"]["ms"] - 4.0) < 1e-9
    # 20 sort launches over 3 epochs: the last 20 // 3 = 6 are charged to the epoch (never all 20)
    assert k["ns::sort_kernel<1>"]["launches_per_epoch"] == 6 and k["ns::sort_kernel<1>"]["launches_total"] == 20
    assert "fill" in d["one_off"] and "fill" not in k
    assert abs(k["ns::walk<4, 4>"]["fetch_bytes"] - 10.0 * 2048) < 1e-6 and abs(k["ns::walk<4, 4>"]["write_bytes"] - 2.0 * 1024) < 1e-6
    assert abs(d["total"]["ms"] - (3.0 + 4.0 + 0.6)) < 1e-6


def test_warp_epoch_row_byte_figures():
    sys.path.insert(0, ROOT)
    import bench
    nnz, d, U, I = 1000, 256, 50, 40
    st = {"accepted": 900, "scored_negatives": 2500, "loaded_rows": 3000, "kernel_ms": 2.0, "aux_ms": 1.0, "optimizer_ms": 0.5}
    r = bench.warp_epoch_row(st, nnz, d, U, I, wall_s=0.004)
    row = 4 * d
    assert r["algorithmic_bytes"] == (8 * 900 + 2 * 100 + 2500) * row + 4 * nnz
    assert r["mean_scored_negatives_T"] == 2.5 and r["candidate_rows_fetched_per_positive"] == 3.0 and r["accepted_frac"] == 0.9
    m = r["implemented_model_bytes"]
    assert m["trial_kernel"] >= (nnz + 3000) * row and m["gather"] >= 2 * 900 * row and m["optimizer"] == (U + I) * 6 * row
    assert abs(r["hbm_frac"] - r["algorithmic_bytes"] / 3e-3 / 1e9 / bench.HBM_PEAK_GBS) < 1e-12   # over trial + sort + gathers
    assert r["implemented_model_total"] == sum(m.values())
