#!/bin/bash
# experiment: K-slice count of the planes bottleneck conv at 8 (video) / 4 (chain) key frames
O=gpurun_out/r06sk; mkdir -p $O
for SK in 0 2 4 6 8; do
  VIDEOSEAL_PLANES_SK=$SK python bench.py --mode video --no-cpu-baseline --steps 20 > $O/video_sk$SK.json 2>/dev/null
done
for SK in 0 3 4 8 12; do
  VIDEOSEAL_PLANES_SK=$SK python bench.py --mode chain --no-cpu-baseline --steps 20 > $O/chain_sk$SK.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline") or {}
        print(f.split("/")[-1], d["value"], d["ms_per_step"], r.get("frac"), r.get("avg_launch_ms"))
    except Exception as e: print(f, "unreadable", e)
PY
