"""Per-kernel summary of rocprofv3 passes of ONE command whose work repeats per epoch: durations from --kernel-trace, HBM bytes
from separate --pmc FETCH_SIZE / WRITE_SIZE passes (MI355X_MICROARCH.md: KiB units; FETCH_SIZE doubled on gfx950 for 16 B/lane
streams).  The launches of every kernel are cut into `epochs` equal groups in dispatch order and the LAST group is reported
(the regime the training settles in), next to the mean over all groups.
    python scripts/pmc_kernels.py <dir with stats/ pmc_fetch/ pmc_write/> <epochs> <out.json> [name-substring ...]"""
import collections
import csv
import glob
import json
import sys

root, epochs, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
want = sys.argv[4:]


def short(k):
    return k.split("(")[0].replace("void ", "").replace("bfh::", "")[:80]


def groups(vals):
    """The launches of the LAST epoch: the last len // epochs of them.  A kernel with fewer launches than epochs ran once per model
    (e.g. the row-id fill, the sort of the static positive list's extra launches) and is reported under `one_off`, not per epoch."""
    n = len(vals) // epochs
    return vals[-n:] if n else []


dur = collections.defaultdict(list)
for f in glob.glob(root + "/stats/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("pmc_fetch", "pmc_write"):
    for f in glob.glob(root + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        if rows and "Dispatch_Id" in rows[0]:
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            cnt[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"epochs": epochs, "kernels": {}, "notes": "ms and bytes PER EPOCH of the last epoch (sum over the kernel's launches in it); "
       "hbm_bytes = 2048 x FETCH_SIZE + 1024 x WRITE_SIZE; memory-side counters include Infinity-Cache hits"}
tot = collections.defaultdict(float)
for k in sorted(set(dur) | set(cnt)):
    if want and not any(w in k for w in want):
        continue
    d_last = groups(dur.get(k, []))
    f_last = groups(cnt[k].get("FETCH_SIZE", []))
    w_last = groups(cnt[k].get("WRITE_SIZE", []))
    if not d_last:
        out.setdefault("one_off", {})[k] = {"launches": len(dur.get(k, [])), "ms": sum(dur.get(k, []))}
        continue
    e = {"launches_per_epoch": len(d_last), "launches_total": len(dur.get(k, [])), "ms": sum(d_last), "fetch_bytes": sum(f_last) * 2048.0,
         "write_bytes": sum(w_last) * 1024.0, "ms_all_epochs_mean": sum(dur.get(k, [])) / max(epochs, 1)}
    e["hbm_bytes"] = e["fetch_bytes"] + e["write_bytes"]
    e["TBps"] = e["hbm_bytes"] / (e["ms"] * 1e-3) / 1e12 if e["ms"] > 0 else None
    out["kernels"][k] = e
    for n in ("ms", "fetch_bytes", "write_bytes", "hbm_bytes"):
        tot[n] += e[n]
out["total"] = dict(tot, TBps=tot["hbm_bytes"] / (tot["ms"] * 1e-3) / 1e12 if tot["ms"] else None, frac_of_8TBps=tot["hbm_bytes"] / (tot["ms"] * 1e-3) / 8e12 if tot["ms"] else None)
json.dump(out, open(out_path, "w"), indent=1)
for k, e in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["ms"])[:12]:
    print("%-60s %3d x  %8.3f ms  %8.3f GB  %s TB/s" % (k[:60], e["launches_per_epoch"], e["ms"], e["hbm_bytes"] / 1e9, "%.2f" % e["TBps"] if e["TBps"] else "-"))
print("total", json.dumps(out["total"]))
