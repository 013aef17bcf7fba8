#!/bin/bash
# round 6, GPU call 4: vectorised resize_pre (+ 4:1 tile), vectorised JPEG, batched GRN fold: tests, fold A/B, the shell rows of the image / chain / ChunkySeal legs
TAG=${1:-r06d}
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_aug.py -m gpu -q > $O/pytest_kernels.log 2>&1
tail -3 $O/pytest_kernels.log
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fwd.py -m gpu -q -x > $O/pytest_e2e.log 2>&1
tail -2 $O/pytest_e2e.log
for i in 1 2 3; do
  VIDEOSEAL_GRN_FOLD=1 python bench.py --no-cpu-baseline --detect-only --steps 40 --warmup 5 --no-kernel-timers > $O/detect_fold$i.json 2>/dev/null
  VIDEOSEAL_GRN_FOLD=0 python bench.py --no-cpu-baseline --detect-only --steps 40 --warmup 5 --no-kernel-timers > $O/detect_nofold$i.json 2>/dev/null
done
python bench.py --no-cpu-baseline --no-extra --steps 20 > $O/image.json 2>/dev/null
python bench.py --no-cpu-baseline --mode chain --steps 20 > $O/chain.json 2>/dev/null
VIDEOSEAL_JPEG=scalar VIDEOSEAL_RESIZE=tile python bench.py --no-cpu-baseline --mode chain --steps 20 > $O/chain_old_forms.json 2>/dev/null
python bench.py --detect-only --card chunkyseal --size 1024 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > $O/chunky.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    sh=(d.get("roofline") or {}).get("shell") or []
    print(f.split("/")[-1], d["value"], d["ms_per_step"], [(x["kernel"][:28], x["frac"], x["avg_launch_ms"]) for x in sh])
PY
ls $O
