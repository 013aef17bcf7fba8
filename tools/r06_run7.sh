#!/bin/bash
# round 6, GPU call 7: ChunkySeal's memory-bound passes (to_planes_affine through LDS, GRN statistics from the GEMM epilogue, dwconv tile shapes)
TAG=${1:-r06g}
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "to_planes or straddl or gemm_planes or dwconv or grn" > $O/pytest_kernels.log 2>&1
tail -3 $O/pytest_kernels.log
for V in auto 0 8 9 10; do
  if [ $V = auto ]; then python tools/bench_dwconv.py chunky > $O/dwconv_chunky_$V.log 2>&1; else VS_DWCONV=$V python tools/bench_dwconv.py chunky > $O/dwconv_chunky_$V.log 2>&1; fi
  grep -E "dwconv7_ln|VS_DWCONV" $O/dwconv_chunky_$V.log
done
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k "chunky or stream or one_group" > $O/pytest_e2e.log 2>&1
tail -2 $O/pytest_e2e.log
python bench.py --detect-only --card chunkyseal --size 1024 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > $O/chunky_new.json 2>/dev/null
VIDEOSEAL_GRN_STRADDLE=0 python bench.py --detect-only --card chunkyseal --size 1024 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > $O/chunky_no_straddle.json 2>/dev/null
python bench.py --mode stream --no-cpu-baseline --steps 3 --warmup 1 > $O/stream.json 2>/dev/null
python bench.py --mode stream --frames 128 --no-cpu-baseline --steps 5 --warmup 2 > $O/stream128.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], d["value"], d["ms_per_step"])
PY
