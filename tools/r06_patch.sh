#!/bin/bash
# conv3x3_patch_kernel: v1 = HEAD, v2 = weight loads unconditional (in-tree), v3 = v2 + branch-free patch prefetch
O=gpurun_out/r06pt; mkdir -p $O
D=$PWD/videoseal_amd/csrc
for m in image video; do
  X=""; [ $m = video ] && X="--mode video"
  for v in v1 v2 v3; do
    L=$D/libvideoseal_$v.so; [ $v = v2 ] && L=$D/libvideoseal_hip.so
    VIDEOSEAL_LIB=$L python bench.py $X --no-cpu-baseline --steps 2 --warmup 1 --no-extra --dump-preds $O/p_${v}_$m.pt > /dev/null 2>&1
  done
  python -c "
import torch; a=torch.load('$O/p_v1_$m.pt'); print('$m identical v2/v3 vs v1:', torch.equal(a, torch.load('$O/p_v2_$m.pt')), torch.equal(a, torch.load('$O/p_v3_$m.pt')))"
done
rm -f $O/*.pt
for i in 1 2 3; do
  for v in v1 v2 v3; do
    L=$D/libvideoseal_$v.so; [ $v = v2 ] && L=$D/libvideoseal_hip.so
    VIDEOSEAL_LIB=$L python bench.py --no-cpu-baseline --steps 30 --warmup 3 --no-kernel-timers --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('image $v', d['ms_per_step'])"
    VIDEOSEAL_LIB=$L python bench.py --mode chain --no-cpu-baseline --steps 30 --warmup 3 --no-kernel-timers --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chain $v', d['ms_per_step'])"
  done
done
cd /tmp && export TMPDIR=/tmp
for v in v1 v2 v3; do
  L=$D/libvideoseal_$v.so; [ $v = v2 ] && L=$D/libvideoseal_hip.so
  VIDEOSEAL_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O -o img_$v -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timers --no-extra --steps 10 --warmup 2 > $GRAFT_REPO_ROOT/$O/img_$v.log 2>&1
  echo "== $v"; grep -E "conv3x3_patch_kernel" $GRAFT_REPO_ROOT/$O/img_${v}_kernel_stats.csv | cut -c1-140
done
rm -f $GRAFT_REPO_ROOT/$O/*_kernel_trace.csv $GRAFT_REPO_ROOT/$O/*agent_info.csv
