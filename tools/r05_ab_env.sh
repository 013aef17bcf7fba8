#!/bin/bash
# Same-box A/B of the round-5 engine switches (the boxes of the pool differ by more than most changes do): every variant twice, alternating.
#   tools/r05_ab_env.sh <tag>   ->  gpurun_out/<tag>/*.json + a table
TAG=${1:-r05ab}
O=gpurun_out/$TAG; mkdir -p $O
run() { # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-cpu-baseline --no-extra "$@" > $O/$name.json 2>$O/$name.err
}
for rep in 1 2; do
  run image_new_$rep X=1 --
  run image_old_$rep VIDEOSEAL_MSG0_PLANES=0 VIDEOSEAL_PW2_NARROW=0 --
  run image_msg0only_$rep VIDEOSEAL_PW2_NARROW=0 --
  run video_new_$rep X=1 -- --mode video
  run video_old_$rep VIDEOSEAL_MSG0_PLANES=0 VIDEOSEAL_PW2_NARROW=0 -- --mode video
  run detect_new_$rep X=1 -- --detect-only --steps 30 --warmup 3
  run detect_old_$rep VIDEOSEAL_PW2_NARROW=0 -- --detect-only --steps 30 --warmup 3
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    r=d.get("roofline") or {}
    print(f.split("/")[-1], d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("frac"))
PY
