"""rocprofv3 --pmc passes of bench.py -> profiles/pmc_latest.json (per launch of the dominant BPR kernel).
usage: python scripts/pmc_summary.py <dir with pmc_fetch/ pmc_write/ [pmc_tcc/]> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from buffalo_amd import _build  # noqa: E402

root, out_path = sys.argv[1], sys.argv[2]
per = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(t in k for t in ("bpr_item_major_kernel", "bpr_item_major_dual_kernel", "bpr_update_kernel", "grad_gather_kernel", "warp_update_kernel", "als_gram_kernel", "als_pc_kernel",
                                    "xcd_merge_kernel", "bpr_presample_kernel")):
            continue
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = {"vgpr": r.get("VGPR_Count"), "sgpr": r.get("SGPR_Count"), "grid": r.get("Grid_Size"), "wg": r.get("Workgroup_Size")}
# the dominant kernel = the one with the largest total FETCH_SIZE (the drain instantiation finds nothing to do)
bpr = [k for k in per if "bpr_item_major_kernel" in k or "bpr_item_major_dual_kernel" in k or "bpr_update_kernel" in k]
dom = max(bpr, key=lambda k: sum(per[k].get("FETCH_SIZE", [0.0])))
c = {n: sum(v) / len(v) for n, v in per[dom].items()}
out = {
    "csrc_sha16": _build.source_fingerprint(),       # bench.py quotes `traffic` from this file only when it runs the same kernel sources
    "command": "rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra "
               "(scripts/gpu_profile.sh; the MFMA-utilisation pass keeps the extras)",
    "kernel": dom, "launches_seen": {n: len(v) for n, v in per[dom].items()}, "counters_per_launch": c, **meta[dom],
    "fetch_bytes_raw": c.get("FETCH_SIZE", 0.0) * 1024, "fetch_bytes_corrected_x2": c.get("FETCH_SIZE", 0.0) * 2048,
    "write_bytes": c.get("WRITE_SIZE", 0.0) * 1024,
    "hbm_bytes_per_launch": c.get("FETCH_SIZE", 0.0) * 2048 + c.get("WRITE_SIZE", 0.0) * 1024,
    "notes": "FETCH_SIZE/WRITE_SIZE are in KiB; separate --pmc passes. FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B "
             "requests at 64 B; calibrated there on 16 B/lane streams, which is what this kernel's float4 row loads are). Memory-side counters "
             "include Infinity-Cache hits. Other kernels of the step (drain instantiation, replica broadcast / merge, pre-sampling) are listed "
             "under 'others'.",
    "others": {k[:90]: {n: sum(v) / len(v) for n, v in d.items()} for k, d in per.items() if k != dom},
}
als = [k for k in per if ("als_pc_kernel" in k or "als_gram_kernel" in k) and "SQ_VALU_MFMA_BUSY_CYCLES" in per[k]]
als.sort(key=lambda k: "als_pc_kernel" not in k)
if als:   # MFMA utilisation of the ALS Gramian/solve kernel: matrix-pipe busy cycles over (kernel cycles x 1024 SIMDs).
    # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (71.4 M for a 3.9 ms launch at ~2.3 GHz), hence the / 8.
    a = {n: sum(v) / len(v) for n, v in per[als[0]].items()}
    out["als_row_kernel"] = {"kernel": als[0][:80], "counters_per_launch": a, **meta[als[0]],
                             "mfma_busy_frac": a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a.get("GRBM_GUI_ACTIVE", 0.0) / 8 * 1024) if a.get("GRBM_GUI_ACTIVE") else None,
                             # v_mfma_f32_32x32x16_f16 keeps the pipe busy 32 cycles (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES = 32 x N for 32x32x16)
                             "mfma_instructions_per_launch_if_32x32x16": a["SQ_VALU_MFMA_BUSY_CYCLES"] / 32.0}
if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
    out["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("kernel", "hbm_bytes_per_launch", "fetch_bytes_corrected_x2", "write_bytes")}))
