#!/usr/bin/env python
"""profiles/<tag>_rocprof_dominant.json and <tag>_pmc_dominant.json from a tools/profile_round.sh output directory: the dominant
kernel's average duration in the rocprofv3 --kernel-trace --stats run of bench.py and its PMC counters (MFMA busy, HBM bytes).
usage: python tools/dominant_json.py gpurun_out/<tag> <tag>      (bench.py reads the newest pair: roofline.frac_rocprof / traffic)"""
import collections, csv, glob, json, os, sys

d, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("conv3x3_pl_kernel<3>", "conv3x3_patch_pc_kernel<3, 8, 2>", "conv3x3_patch_pc_kernel<3, 8, 3>")
rows = list(csv.DictReader(open(os.path.join(d, "img_kernel_stats.csv"))))
row = next(r for k in KEYS for r in rows if k in r["Name"])
kern = next(k for k in KEYS if k in row["Name"])
arith = 3 if kern.endswith("3>") and "patch_pc" in kern else 2
src = f"profiles/{tag}_bench_image_b32_768_kernel_stats.csv (rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-kernel-timers --steps 10 --warmup 2)"
json.dump({"kernel": kern, "arith": arith, "avg_ms": round(float(row["AverageNs"]) / 1e6, 4), "calls": int(row["Calls"]), "source": src},
          open(os.path.join(ROOT, "profiles", f"{tag}_rocprof_dominant.json"), "w"))
agg = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(d, "pmc", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
if agg:
    mf, gui = agg.get("SQ_VALU_MFMA_BUSY_CYCLES"), agg.get("GRBM_GUI_ACTIVE")
    out = {"kernel": kern, "arith": arith,
           "fetch_mb_per_launch": round(2.0 * sum(agg["FETCH_SIZE"]) / len(agg["FETCH_SIZE"]) / 1024, 1) if agg.get("FETCH_SIZE") else None,
           "write_mb_per_launch": round(sum(agg["WRITE_SIZE"]) / len(agg["WRITE_SIZE"]) / 1024, 1) if agg.get("WRITE_SIZE") else None,
           "mfma_busy_pct": round(100.0 * sum(mf) / (sum(gui) / 8.0 * 1024), 1) if mf and gui else None,
           # planes in (B*H*W*C*4 B) + weights (N*K*2 planes*2 B) + planes out; the fused 1x1 launches read a second planes tensor
           "algorithmic_mb_per_launch": round((2 * 32768 * 384 * 4 + 384 * 3456 * 4) / 1e6, 1) if "pl_kernel" in kern else 108.6,
           "source": f"profiles/{tag}_pmc_summary.md (rocprofv3 --pmc FETCH_SIZE x2 gfx950 correction, WRITE_SIZE; separate passes)"}
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_dominant.json"), "w"))
print(open(os.path.join(ROOT, "profiles", f"{tag}_rocprof_dominant.json")).read())
