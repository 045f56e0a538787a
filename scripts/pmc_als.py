"""Mean hardware-counter values per dispatch of the ALS row kernel from rocprofv3 --pmc passes (one directory per pass):
    python scripts/pmc_als.py DIR [DIR ...]"""
import csv, glob, os, sys
from collections import defaultdict
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(list)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if "als_gram_kernel" not in name:
                    continue
                split = "split" if name.rstrip(">) ").endswith("true") or "Lb1EEE" in name or ", true>" in name else "fp32"
                acc[(split, row["Counter_Name"], row.get("Grid_Size", ""))].append(float(row["Counter_Value"]))
        for (split, c, g), v in sorted(acc.items()):
            print("%-6s grid %-8s %-28s mean %.4g  (n=%d; first %.4g last %.4g)" % (split, g, c, sum(v) / len(v), len(v), v[0], v[-1]))
