"""Per-launch counters of the ALS row kernel (als_pc_kernel, configs[2]) from a rocprofv3 --kernel-trace --pmc pass of scripts/als_extra_only.py:
user-side and item-side launches apart (by grid order), mean duration and counters.  usage: r6_als_pmc.py <dir>"""
import collections, csv, glob, sys
root = sys.argv[1]
dur, grid = {}, {}
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "als_pc_kernel" in r["Kernel_Name"]:
            dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
cnt = collections.defaultdict(dict)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "als_pc_kernel" in r["Kernel_Name"]:
            cnt[int(r["Dispatch_Id"])][r["Counter_Name"]] = cnt[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(set(dur) | set(cnt))
names = sorted({n for d in cnt.values() for n in d})
print("launches seen: %d; counters: %s" % (len(ids), names))
for side, sel in (("even launches (first half-epoch of a pair)", ids[0::2]), ("odd launches", ids[1::2])):
    sel = sel[1:]   # past the first epoch
    d = [dur[i] for i in sel if i in dur]
    line = "%s: n=%d %.3f ms" % (side, len(sel), sum(d) / max(1, len(d)))
    for n in names:
        v = [cnt[i][n] for i in sel if n in cnt.get(i, {})]
        line += "  %s %.4g" % (n, sum(v) / max(1, len(v)))
    print(line)
