#!/bin/bash
# SQ issue / wait counters of every kernel whose name contains <substring>, averaged per launch, for an arbitrary command.
# usage: tools/pmc_kernel.sh <outdir> <substring> <command...>   (GPU box; counter passes only -- never together with trace flags)
R=$PWD; O=$R/$1; SUB=$2; shift 2; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA"; do
  N=$(echo $P | cut -d' ' -f1)
  (cd $R && timeout 300 rocprofv3 --pmc $P --output-format csv -d $O -o $N -- "$@" > $O/$N.log 2>&1)
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "$SUB" not in kn: continue
        k = kn.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in sorted(agg.items()):
    d = {n: round(sum(v) / len(v)) for n, v in sorted(c.items())}
    wc = d.get("SQ_WAVE_CYCLES") or 1
    d["derived"] = {"launches": len(next(iter(c.values()))), "valu_per_mfma": round(d.get("SQ_INSTS_VALU", 0) / max(1, d.get("SQ_INSTS_MFMA", 1)), 1),
                    "wait_inst_frac_of_wave_cycles": round(d.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                    "active_inst_frac_of_wave_cycles": round(d.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
                    "lds_bank_conflict_frac_of_lds_active": round(d.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, d.get("SQ_LDS_IDX_ACTIVE", 1)), 3)}
    out[k] = d
json.dump(out, open("$O/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
