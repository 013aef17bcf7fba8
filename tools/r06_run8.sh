#!/bin/bash
# round 6, GPU call 8: ChunkySeal with the planes pwconv2 on its odd-frame stages + batched straddle finish; full GPU suite on the tree
TAG=${1:-r06h}
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
python bench.py --detect-only --card chunkyseal --size 1024 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > $O/chunky.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("$O/chunky.json").read().strip().splitlines()[-1]); print("chunky", d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("e2e_frac"))
PY
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_zdist.py::test_bench_eight_ranks_on_one_gpu_preflight > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o chunky -- python $R/bench.py --detect-only --card chunkyseal --size 1024 --batch 16 --steps 12 --warmup 2 --no-cpu-baseline --no-kernel-timers > $O/chunky_prof.log 2>&1
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv
