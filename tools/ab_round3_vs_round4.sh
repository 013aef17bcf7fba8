mkdir -p gpurun_out/r04j; O=gpurun_out/r04j
for m in image video chain; do
  extra=""; [ $m != image ] && extra="--mode $m"
  python bench.py --no-cpu-baseline --no-extra $extra > $O/${m}_new.json 2>/dev/null
  VIDEOSEAL_THIN_FUSED=0 VIDEOSEAL_TAIL=sep VIDEOSEAL_RESIZE=tile python bench.py --no-cpu-baseline --no-extra $extra > $O/${m}_r3.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-extra $extra > $O/${m}_new2.json 2>/dev/null
done
python bench.py --no-cpu-baseline --mode stream --steps 3 --warmup 1 > $O/stream_new.json 2>/dev/null
VIDEOSEAL_THIN_FUSED=0 VIDEOSEAL_TAIL=sep VIDEOSEAL_RESIZE=tile python bench.py --no-cpu-baseline --mode stream --group 1 --steps 3 --warmup 1 > $O/stream_r3.json 2>/dev/null
python bench.py --no-cpu-baseline --mode stream --no-overlap --steps 3 --warmup 1 > $O/stream_new_seq_calls.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r04j/*.json")):
    d=json.load(open(f)); r=d.get("roofline") or {}
    print(f.split("/")[-1], d["value"], d["ms_per_step"], r.get("frac"), [(s["kernel"], s["avg_launch_ms"]) for s in (r.get("shell") or [])][:2])
PY
