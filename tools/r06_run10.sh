#!/bin/bash
# round 6, GPU call 10: the 256 x 256 one-wave-per-SIMD planes GEMM (tile 27): parity with tile 24, ChunkySeal A/B
TAG=${1:-r06j}
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "big_tile or dwconv7_ln_two or gemm_planes" > $O/pytest_kernels.log 2>&1
tail -3 $O/pytest_kernels.log
for i in 1 2; do
VIDEOSEAL_GEMM_BIG=1 timeout 300 python bench.py --detect-only --card chunkyseal --size 1024 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > $O/chunky_big$i.json 2>$O/chunky_big$i.err
VIDEOSEAL_GEMM_BIG=0 timeout 300 python bench.py --detect-only --card chunkyseal --size 1024 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > $O/chunky_t24_$i.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/chunky*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("avg_launch_ms"))
    except Exception as e: print(f, "unreadable", e)
PY
tail -3 $O/chunky_big1.err
