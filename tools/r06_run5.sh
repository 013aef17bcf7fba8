#!/bin/bash
# round 6, GPU call 5: shared row-streaming resize (vertical weight table, crop windows, aug epilogue): tests + shell rows of the legs
TAG=${1:-r06e}
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_aug.py -m gpu -q > $O/pytest_kernels.log 2>&1
tail -3 $O/pytest_kernels.log
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fwd.py tests/test_gpu_shim.py -m gpu -q -x > $O/pytest_e2e.log 2>&1
tail -2 $O/pytest_e2e.log
python bench.py --no-cpu-baseline --no-extra --steps 20 > $O/image.json 2>/dev/null
VIDEOSEAL_RESIZE=tile python bench.py --no-cpu-baseline --no-extra --steps 20 > $O/image_tile_resize.json 2>/dev/null
python bench.py --no-cpu-baseline --mode chain --steps 20 > $O/chain.json 2>/dev/null
VIDEOSEAL_CROP_RESIZE=tile python bench.py --no-cpu-baseline --mode chain --steps 20 > $O/chain_tile_crop_resize.json 2>/dev/null
python bench.py --no-cpu-baseline --mode video --steps 20 > $O/video.json 2>/dev/null
python bench.py --detect-only --card chunkyseal --size 1024 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > $O/chunky.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    sh=(d.get("roofline") or {}).get("shell") or []
    print(f.split("/")[-1], d["value"], d["ms_per_step"], [(x["kernel"][:28], x["frac"], x["avg_launch_ms"]) for x in sh])
PY
