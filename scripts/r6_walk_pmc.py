"""Per-HANDLE counters of the headline kernel: a rocprofv3 --kernel-trace --pmc pass of scripts/r6_walk_variance.py (REPS handles, 45 epochs = 90
launches each) -> per handle the mean launch duration (kernel trace) and the mean of every counter collected.  usage: r6_walk_pmc.py <dir> [launches per handle]"""
import collections, csv, glob, sys
root = sys.argv[1]
per_handle = int(sys.argv[2]) if len(sys.argv) > 2 else 90
dur = {}
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bpr_item_major_dual_kernel" in r["Kernel_Name"]:
            dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
cnt = collections.defaultdict(dict)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bpr_item_major_dual_kernel" in r["Kernel_Name"]:
            cnt[int(r["Dispatch_Id"])][r["Counter_Name"]] = cnt[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(set(dur) | set(cnt))
names = sorted({n for d in cnt.values() for n in d})
print("launches seen: %d; counters: %s" % (len(ids), names))
for h in range(0, len(ids), per_handle):
    grp = ids[h + 10:h + per_handle]      # past the handle's warm-up epochs
    if not grp:
        break
    d = [dur[i] for i in grp if i in dur]
    line = "handle %d: %.3f ms (min %.3f max %.3f)" % (h // per_handle, sum(d) / max(1, len(d)), min(d or [0]), max(d or [0]))
    for n in names:
        v = [cnt[i][n] for i in grp if n in cnt.get(i, {})]
        line += "  %s %.4g" % (n, sum(v) / max(1, len(v)))
    print(line)
