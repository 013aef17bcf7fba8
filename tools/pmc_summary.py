#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (counter_collection CSVs) per kernel: MFMA utilisation and HBM traffic.
usage: python tools/pmc_summary.py <dir with pN_counter_collection.csv> > profiles/<name>.md"""
import collections, csv, glob, os, re, sys

d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        k = re.sub(r"\((vs_conv_desc|float|unsigned|TailArgs|at::|int|long).*", "", k)
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["_dur_ns:" + os.path.basename(f) + ":" + r["Counter_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("| kernel | launches | avg ms | MFMA busy % of SIMD cycles | FETCH_SIZE MB/launch (x2 corrected) | WRITE_SIZE MB/launch | HBM GB/s |")
print("|---|---|---|---|---|---|---|")
rows = []
for k, c in agg.items():
    durs = [v for kk, v in c.items() if kk.startswith("_dur_ns")]
    n = len(durs[0])          # dispatches of this kernel in one pass
    avg_ms = sum(sum(v) for v in durs) / sum(len(v) for v in durs) / 1e6
    mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES"); gui = c.get("GRBM_GUI_ACTIVE")
    util = ""
    if mfma and gui:   # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
        util = f"{100.0 * sum(mfma) / (sum(gui) / 8.0 * 1024):.1f}"
    fetch = c.get("FETCH_SIZE"); wr = c.get("WRITE_SIZE")
    fmb = (2.0 * sum(fetch) / len(fetch) / 1024) if fetch else None      # KB -> MB, x2 gfx950 correction (MI355X_MICROARCH.md HBM)
    wmb = (sum(wr) / len(wr) / 1024) if wr else None
    bw = f"{((fmb or 0) + (wmb or 0)) / avg_ms:.0f}" if (fmb is not None or wmb is not None) else ""
    rows.append((avg_ms * n, f"| `{k}` | {n} | {avg_ms:.3f} | {util} | {'' if fmb is None else f'{fmb:.1f}'} | {'' if wmb is None else f'{wmb:.1f}'} | {bw} |"))
for _, line in sorted(rows, reverse=True)[:18]:
    print(line)
