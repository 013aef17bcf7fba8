#!/bin/bash
O=gpurun_out/r06rb; mkdir -p $O
PREV=$PWD/videoseal_amd/csrc/libvideoseal_prev.so
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "resblock or thin or unet" 2>&1 | tail -1
for m in image video; do
  X=""; [ $m = video ] && X="--mode video"
  python bench.py $X --no-cpu-baseline --steps 2 --warmup 1 --no-extra --dump-preds $O/p_new_$m.pt > /dev/null 2>&1
  VIDEOSEAL_LIB=$PREV python bench.py $X --no-cpu-baseline --steps 2 --warmup 1 --no-extra --dump-preds $O/p_prev_$m.pt > /dev/null 2>&1
  python -c "
import torch; print('$m preds identical:', torch.equal(torch.load('$O/p_new_$m.pt'), torch.load('$O/p_prev_$m.pt')))"
done
rm -f $O/*.pt
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 30 --warmup 3 --no-kernel-timers --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('image new', d['ms_per_step'])"
  VIDEOSEAL_LIB=$PREV python bench.py --no-cpu-baseline --steps 30 --warmup 3 --no-kernel-timers --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('image prev (before patch + upconv changes)', d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
for v in new prev; do
  L=$GRAFT_REPO_ROOT/videoseal_amd/csrc/libvideoseal_hip.so; [ $v = prev ] && L=$PREV
  VIDEOSEAL_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O -o img_$v -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timers --no-extra --steps 10 --warmup 2 > $GRAFT_REPO_ROOT/$O/img_$v.log 2>&1
  echo "== $v"; grep -E "resblock_thin|upconv_fused|conv3x3_patch_kernel" $GRAFT_REPO_ROOT/$O/img_${v}_kernel_stats.csv | cut -c1-150
done
rm -f $GRAFT_REPO_ROOT/$O/*_kernel_trace.csv $GRAFT_REPO_ROOT/$O/*agent_info.csv
