#!/bin/bash
# round 6, GPU call 9: two-row strips of the one-row depthwise kernel (ChunkySeal), wider batches in the straddle finish
TAG=${1:-r06i}
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "dwconv or straddl" > $O/pytest_kernels.log 2>&1
tail -2 $O/pytest_kernels.log
python tools/bench_dwconv.py chunky 2>&1 | grep dwconv7_ln
VS_DWCONV_ROWS=1 python tools/bench_dwconv.py chunky 2>&1 | grep dwconv7_ln
VS_DWCONV_ROWS=2 python tools/bench_dwconv.py 2>&1 | grep dwconv7_ln
VS_DWCONV_ROWS=1 python tools/bench_dwconv.py 2>&1 | grep dwconv7_ln
python bench.py --detect-only --card chunkyseal --size 1024 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > $O/chunky.json 2>/dev/null
VS_DWCONV_ROWS=1 python bench.py --detect-only --card chunkyseal --size 1024 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > $O/chunky_rows1.json 2>/dev/null
python - <<PY
import json
for n in ("chunky","chunky_rows1"):
    d=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1]); print(n, d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("e2e_frac"))
PY
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k "chunky" > $O/pytest_e2e.log 2>&1
tail -2 $O/pytest_e2e.log
