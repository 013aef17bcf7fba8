#!/bin/bash
# SQ issue / wait counters of the fused ConvNeXt block kernel (tools/bench_cnx.py).  usage: tools/pmc_cnx.sh <outdir>   (GPU box)
R=$PWD; O=$R/${1:-gpurun_out/pmc_cnx}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA"; do
  N=$(echo $P | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $P --output-format csv -d $O -o $N -- python $R/tools/bench_cnx.py > $O/$N.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "cnx_block_kernel" not in kn: continue
        k = kn[kn.index("cnx_block_kernel"):kn.index(">") + 1]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(agg.items()):
    print(k, {n: round(sum(v) / len(v)) for n, v in sorted(c.items())})
PY
