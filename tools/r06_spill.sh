#!/bin/bash
# spill removal in the TN = 3 GEMM tiles: kernel + end-to-end tests on the new build, then a same-box A/B against the previous build
# (videoseal_amd/csrc/libvideoseal_prev.so, built from the parent commit) on the detect-only, image and ChunkySeal lines + bit-identity of the decisions
O=gpurun_out/r06pk; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > $O/pytest_kernels.log 2>&1; tail -3 $O/pytest_kernels.log
PREV=$PWD/videoseal_amd/csrc/libvideoseal_prev.so
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --detect-only --steps 40 --warmup 5 --no-kernel-timers --no-extra > $O/detect_new$i.json 2>/dev/null
  VIDEOSEAL_LIB=$PREV python bench.py --no-cpu-baseline --detect-only --steps 40 --warmup 5 --no-kernel-timers --no-extra > $O/detect_prev$i.json 2>/dev/null
done
for i in 1 2; do
  python bench.py --no-cpu-baseline --steps 20 --warmup 3 --no-extra > $O/image_new$i.json 2>/dev/null
  VIDEOSEAL_LIB=$PREV python bench.py --no-cpu-baseline --steps 20 --warmup 3 --no-extra > $O/image_prev$i.json 2>/dev/null
  python bench.py --no-cpu-baseline --card chunkyseal --size 1024 --batch 16 --detect-only --steps 3 --warmup 1 --no-extra --no-kernel-timers > $O/chunky_new$i.json 2>/dev/null
  VIDEOSEAL_LIB=$PREV python bench.py --no-cpu-baseline --card chunkyseal --size 1024 --batch 16 --detect-only --steps 3 --warmup 1 --no-extra --no-kernel-timers > $O/chunky_prev$i.json 2>/dev/null
done
python bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-extra --dump-preds $O/preds_new.pt > /dev/null 2>&1
VIDEOSEAL_LIB=$PREV python bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-extra --dump-preds $O/preds_prev.pt > /dev/null 2>&1
python - <<PY
import json,glob,torch
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline") or {}
        print(f.split("/")[-1], d["value"], d["ms_per_step"], r.get("frac"))
    except Exception as e: print(f, "unreadable", e)
try:
    a=torch.load("$O/preds_new.pt"); b=torch.load("$O/preds_prev.pt")
    print("preds identical:", all(torch.equal(x,y) for x,y in zip(a,b)) if isinstance(a,(list,tuple)) else torch.equal(a,b))
except Exception as e: print("preds compare failed", e)
PY
rm -f $O/preds_new.pt $O/preds_prev.pt
timeout 1800 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fwd.py tests/test_gpu_shim.py -m gpu -q -x > $O/pytest_e2e.log 2>&1; tail -2 $O/pytest_e2e.log
