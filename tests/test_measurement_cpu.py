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
    assert abs(r["survey_formula_bytes_over_peak"] - r["algorithmic_bytes"] / 3e-3 / 1e9 / bench.HBM_PEAK_GBS) < 1e-12   # over trial + sort + gathers
    assert r["implemented_model_total"] == sum(m.values())


def test_bench_last_line_is_compact_and_keeps_the_contract():
    """BENCH_r04.json.parsed was null: the 27 KB line did not fit the driver's 8 KB stdout tail.  The line-assembly code run on that very
    line (profiles/r04_bench_n1_default_250_steps.json, a real full result) must give <= 4096 bytes that json.loads back with the
    contract's head keys, the flat roofline (frac, kernel_ms, the ALS / WARP keys) and the flat cpu_baseline."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_n1_default_250_steps.json")))
    assert len(json.dumps(full)) > 20000                                   # the canned input is the oversized one
    full["config"].update({"lr": 0.002, "min_lr": 0.0001, "num_negative_samples": 1})
    full["roofline"].update({"bpr_lr005_kernel_ms": 5.1, "bpr_lr005_frac": 0.76, "als_user_max": 1.2e-4, "als_item_max": 1.7e-2,
                             "als_item_max_reorder": 2.4e-2, "als_top10_overlap": 0.9755, "als_top10_overlap_reorder": 0.9757,
                             "traffic_frac": 0.62})
    full["cpu_baseline"].update({"reference_on_stand_ins_value": 5.9e6, "reference_on_stand_ins_kind": "reference-on-stand-ins"})
    s = bench.compact_line(full)
    assert len(s) <= bench.LINE_LIMIT and "\n" not in s
    line = json.loads(s)
    for k in bench.HEAD_KEYS:
        assert k in line, k
    assert line["value"] == float("%.6g" % full["value"]) and line["n_gpus"] == 1 and line["higher_is_better"] is True
    rf, cb = line["roofline"], line["cpu_baseline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "als_epoch_ms", "als_kernel_ms", "als_hbm_frac",
              "warp_c5_epoch_ms", "warp_c5_T", "warp_ml20m_epoch_ms", "bpr_lr005_kernel_ms", "bpr_lr005_frac", "als_user_max", "als_top10_overlap",
              "triad_GBps", "frac_of_triad"):
        assert k in rf, k
    assert all(not isinstance(v, (dict, list)) for v in rf.values()) and all(not isinstance(v, (dict, list)) for v in cb.values())
    assert abs(rf["frac"] - full["roofline"]["frac"]) < 1e-5 and abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-5
    for k in ("value", "unit", "cores", "kind", "sample", "als_value", "warp_c5_value", "reference_on_stand_ins_value"):
        assert k in cb, k
    assert line["config"]["lr"] == 0.002 and "workload" in line["config"]
    # a pathological result (every string long, hundreds of flat keys) still fits: optional keys go first, the contract's keys never
    fat = json.loads(json.dumps(full))
    fat["roofline"].update({"zz_%d" % i: 1.0 / (i + 1) for i in range(40)})
    fat["config"]["workload"] = "x" * 3000
    fat["cpu_baseline"]["sample"] = "y" * 3000
    s2 = bench.compact_line(fat)
    assert len(s2) <= bench.LINE_LIMIT
    l2 = json.loads(s2)
    assert "frac" in l2["roofline"] and "value" in l2["cpu_baseline"] and l2["value"] == line["value"]
    # N > 1 (round 6): what the LIVE communicator reports travels as top-level scalars of the line; the round-6 keys of the N = 1 line are flat scalars too
    multi = json.loads(json.dumps(full))
    multi.update({"n_gpus": 8, "rccl_ranks": 8, "transport": "rccl 2.26.6"})
    multi["roofline"].update({"bpr_lr005_norm_gap_Qb": -0.0375, "warp_c5_agrees_with_device": True, "warp_c5_search_T": 23.7, "warp_c5_search_epoch_ms": 4257.7,
                              "warp_c5_search_accepted_frac": 0.9967})
    l3 = json.loads(bench.compact_line(multi))
    assert l3["rccl_ranks"] == 8 and l3["transport"] == "rccl 2.26.6" and l3["n_gpus"] == 8
    assert l3["roofline"]["warp_c5_agrees_with_device"] is True and l3["roofline"]["bpr_lr005_norm_gap_Qb"] == -0.0375
    assert "rccl_ranks" not in line and "transport" not in line          # ... and are absent at N = 1


def test_bench_counter_traffic_is_refused_from_other_sources(tmp_path, monkeypatch):
    """`roofline.traffic` comes from a file an EARLIER process wrote: it is quoted only when that process ran these kernel sources."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    from buffalo_amd import _build
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (prof / "pmc_latest.json").write_text(json.dumps({"hbm_bytes_per_launch": 2.0e10, "csrc_sha16": "0123456789abcdef"}))
    v, why = bench.counter_traffic()
    assert v is None and "other kernel sources" in why
    (prof / "pmc_latest.json").write_text(json.dumps({"hbm_bytes_per_launch": 2.0e10, "csrc_sha16": _build.source_fingerprint()}))
    v, why = bench.counter_traffic()
    assert v == 2.0e10
    (prof / "pmc_latest.json").unlink()
    assert bench.counter_traffic()[0] is None


def test_bench_self_launch_sets_the_rendezvous(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts N ranks of itself with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set
    (the driver's torchrun form sets WORLD_SIZE and never takes this branch); a dying rank takes the others down and its code is returned."""
    sys.path.insert(0, ROOT)
    import subprocess
    import bench
    seen = []

    class P:
        def __init__(self, cmd, env=None, stdout=None):
            seen.append((cmd, env, stdout))
            self.rc = 3 if env["RANK"] == "1" else None
            self.terminated = False

        def poll(self):
            return self.rc

        def terminate(self):
            self.terminated, self.rc = True, -15
    monkeypatch.setattr(subprocess, "Popen", P)
    args = bench.parse_args(["--gpus", "4", "--workload", "warp_c5"])
    rc = bench.self_launch(args, ["--gpus", "4", "--workload", "warp_c5"])
    assert rc == 3 and len(seen) == 4
    ports = {e["MASTER_PORT"] for _, e, _ in seen}
    assert len(ports) == 1 and all(e["MASTER_ADDR"] == "127.0.0.1" and e["WORLD_SIZE"] == "4" for _, e, _ in seen)
    assert [e["RANK"] for _, e, _ in seen] == ["0", "1", "2", "3"] == [e["LOCAL_RANK"] for _, e, _ in seen]
    assert all(c[-4:] == ["--gpus", "4", "--workload", "warp_c5"] and c[1].endswith("bench.py") for c, _, _ in seen)
    assert seen[0][2] is None and seen[1][2] is not None                 # rank 0 owns stdout
