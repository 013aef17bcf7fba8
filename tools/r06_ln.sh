#!/bin/bash
O=gpurun_out/r06ln; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "layernorm" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --detect-only --steps 40 --warmup 5 --no-kernel-timers > $O/detect_lanes$i.json 2>/dev/null
  VIDEOSEAL_LN=wave python bench.py --no-cpu-baseline --detect-only --steps 40 --warmup 5 --no-kernel-timers > $O/detect_wave$i.json 2>/dev/null
done
python bench.py --detect-only --card chunkyseal --size 1024 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > $O/chunky.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"])
PY
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fwd.py -m gpu -q -x > $O/pytest_e2e.log 2>&1; tail -2 $O/pytest_e2e.log
