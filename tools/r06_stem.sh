#!/bin/bash
O=gpurun_out/r06st; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "stem or layernorm" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --detect-only --steps 40 --warmup 5 --no-kernel-timers > $O/detect_fused$i.json 2>/dev/null
  VIDEOSEAL_STEM_FUSED=0 python bench.py --no-cpu-baseline --detect-only --steps 40 --warmup 5 --no-kernel-timers > $O/detect_two$i.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"])
PY
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fwd.py tests/test_gpu_shim.py -m gpu -q -x > $O/pytest_e2e.log 2>&1; tail -2 $O/pytest_e2e.log
