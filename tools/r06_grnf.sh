#!/bin/bash
O=gpurun_out/r06gf; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "grn or cnx or convnext" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
PREV=$PWD/videoseal_amd/csrc/libvideoseal_prev.so
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --detect-only --steps 40 --warmup 5 --no-kernel-timers --no-extra > $O/detect_new$i.json 2>/dev/null
  VIDEOSEAL_LIB=$PREV python bench.py --no-cpu-baseline --detect-only --steps 40 --warmup 5 --no-kernel-timers --no-extra > $O/detect_prev$i.json 2>/dev/null
done
python bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-extra --dump-preds $O/p_new.pt > /dev/null 2>&1
VIDEOSEAL_LIB=$PREV python bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-extra --dump-preds $O/p_prev.pt > /dev/null 2>&1
python - <<PY
import json,glob,torch
for f in sorted(glob.glob("$O/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"])
print("preds identical:", torch.equal(torch.load("$O/p_new.pt"), torch.load("$O/p_prev.pt")))
PY
rm -f $O/p_new.pt $O/p_prev.pt
timeout 1200 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x > $O/pytest_e2e.log 2>&1; tail -1 $O/pytest_e2e.log
