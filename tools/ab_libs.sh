#!/bin/bash
# Same-box A/B of two builds of the library (the boxes of the pool differ by more than most changes do).
#   here:   git archive <ref> videoseal_amd/csrc include | tar -x -C /tmp/ab_ref; make -C /tmp/ab_ref/videoseal_amd/csrc
#           cp /tmp/ab_ref/videoseal_amd/csrc/libvideoseal_hip.so videoseal_amd/csrc/build_ab/libvideoseal_hip_ref.so     (git-ignored, travels with gpurun)
#   on box: tools/ab_libs.sh <tag>      ->  gpurun_out/<tag>/{image,video,chain,stream,detect}_{new,ref}*.json + a table
TAG=${1:-ab}
O=gpurun_out/$TAG; mkdir -p $O
REF=$PWD/videoseal_amd/csrc/build_ab/libvideoseal_hip_ref.so
[ -f $REF ] || { echo "no $REF"; exit 1; }
for m in image video chain; do
  extra=""; [ $m != image ] && extra="--mode $m"
  python bench.py --no-cpu-baseline --no-extra $extra > $O/${m}_new.json 2>/dev/null
  VIDEOSEAL_LIB=$REF python bench.py --no-cpu-baseline --no-extra $extra > $O/${m}_ref.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-extra $extra > $O/${m}_new2.json 2>/dev/null
  VIDEOSEAL_LIB=$REF python bench.py --no-cpu-baseline --no-extra $extra > $O/${m}_ref2.json 2>/dev/null
done
python bench.py --no-cpu-baseline --mode stream --steps 3 --warmup 1 > $O/stream_new.json 2>/dev/null
VIDEOSEAL_LIB=$REF python bench.py --no-cpu-baseline --mode stream --steps 3 --warmup 1 > $O/stream_ref.json 2>/dev/null
python bench.py --no-cpu-baseline --detect-only --steps 30 --warmup 3 > $O/detect_new.json 2>/dev/null
VIDEOSEAL_LIB=$REF python bench.py --no-cpu-baseline --detect-only --steps 30 --warmup 3 > $O/detect_ref.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    r=d.get("roofline") or {}
    print(f.split("/")[-1], d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("frac"))
PY
