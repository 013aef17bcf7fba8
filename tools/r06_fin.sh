#!/bin/bash
# K slices added inside the wave-specialised GEMM (csrc/splitk_finish.h): parity test, then same-box A/B of the chain / video / detect(16) lines
O=gpurun_out/r06fin; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "k_slices_added or gemm1x1_pc or wave_specialised_gemm" > $O/pytest_fin.log 2>&1; tail -3 $O/pytest_fin.log
for i in 1 2 3; do
  for F in 0 1; do
    VIDEOSEAL_SPLITK_FINISH=$F python bench.py --mode chain --no-cpu-baseline --steps 30 --warmup 3 --no-kernel-timers --no-extra > $O/chain_f${F}_$i.json 2>/dev/null
    VIDEOSEAL_SPLITK_FINISH=$F python bench.py --no-cpu-baseline --detect-only --batch 16 --steps 40 --warmup 5 --no-kernel-timers --no-extra > $O/det16_f${F}_$i.json 2>/dev/null
  done
done
for F in 0 1; do
  VIDEOSEAL_SPLITK_FINISH=$F python bench.py --mode video --no-cpu-baseline --steps 30 --warmup 3 --no-kernel-timers --no-extra > $O/video_f${F}.json 2>/dev/null
  VIDEOSEAL_SPLITK_FINISH=$F python bench.py --mode chain --no-cpu-baseline --steps 2 --warmup 1 --no-extra --dump-preds $O/preds_f$F.pt > /dev/null 2>&1
done
python - <<PY
import json,glob,torch
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"])
    except Exception as e: print(f, "unreadable", e)
try:
    a=torch.load("$O/preds_f0.pt"); b=torch.load("$O/preds_f1.pt"); print("chain preds identical:", torch.equal(a,b))
except Exception as e: print("preds compare failed", e)
PY
rm -f $O/preds_f0.pt $O/preds_f1.pt
cd /tmp && export TMPDIR=/tmp
VIDEOSEAL_SPLITK_FINISH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O -o chain_fin -- python $GRAFT_REPO_ROOT/bench.py --mode chain --no-cpu-baseline --no-kernel-timers --no-extra --steps 10 --warmup 2 > $GRAFT_REPO_ROOT/$O/chain_fin.log 2>&1
rm -f $GRAFT_REPO_ROOT/$O/*_kernel_trace.csv $GRAFT_REPO_ROOT/$O/*agent_info.csv
