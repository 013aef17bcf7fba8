#!/bin/bash
# per-kernel same-box comparison of two library builds under rocprofv3 (image mode): which kernels moved, not just the step
#   tools/ab_kernel_stats.sh <tag>   (reference library: videoseal_amd/csrc/build_ab/libvideoseal_hip_ref.so)
TAG=${1:-abk}; O=$PWD/gpurun_out/$TAG; mkdir -p $O
REF=$PWD/videoseal_amd/csrc/build_ab/libvideoseal_hip_ref.so
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in new ref; do
  L=""; [ $v = ref ] && L=$REF
  VIDEOSEAL_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python $R/bench.py --no-cpu-baseline --no-extra ${BENCH_ARGS} > $O/bench_$v.json 2>/dev/null
  cp $(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$v.csv
done
python - <<PY
import csv
def load(f): return {r['Name']:(int(r['Calls']),float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/1e6) for r in csv.DictReader(open(f))}
a=load("$O/kernel_stats_new.csv"); b=load("$O/kernel_stats_ref.csv")
rows=[]
for k in set(a)|set(b):
    ca,ua,ta=a.get(k,(0,0,0)); cb,ub,tb=b.get(k,(0,0,0))
    rows.append((ta-tb,k,ca,ua,cb,ub))
rows.sort(key=lambda r:-abs(r[0]))
print("delta_total_ms(new-ref)  calls_new avg_us_new | calls_ref avg_us_ref  kernel")
for d,k,ca,ua,cb,ub in rows[:25]: print(f"{d:+8.3f}  {ca:5d} {ua:8.1f} | {cb:5d} {ub:8.1f}  {k[:90]}")
PY
