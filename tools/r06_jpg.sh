#!/bin/bash
O=gpurun_out/r06jp; mkdir -p $O
PREV=$PWD/videoseal_amd/csrc/libvideoseal_prev.so
timeout 900 python -m pytest tests/test_gpu_aug.py -m gpu -q -x 2>&1 | tail -1
python bench.py --mode chain --no-cpu-baseline --steps 2 --warmup 1 --no-extra --dump-preds $O/p_new.pt > /dev/null 2>&1
VIDEOSEAL_LIB=$PREV python bench.py --mode chain --no-cpu-baseline --steps 2 --warmup 1 --no-extra --dump-preds $O/p_prev.pt > /dev/null 2>&1
python -c "
import torch; print('chain preds identical:', torch.equal(torch.load('$O/p_new.pt'), torch.load('$O/p_prev.pt')))"
rm -f $O/*.pt
for i in 1 2 3; do
  python bench.py --mode chain --no-cpu-baseline --steps 30 --warmup 3 --no-kernel-timers --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chain new', d['ms_per_step'])"
  VIDEOSEAL_LIB=$PREV python bench.py --mode chain --no-cpu-baseline --steps 30 --warmup 3 --no-kernel-timers --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chain prev', d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
for v in new prev; do
  L=$GRAFT_REPO_ROOT/videoseal_amd/csrc/libvideoseal_hip.so; [ $v = prev ] && L=$PREV
  VIDEOSEAL_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O -o ch_$v -- python $GRAFT_REPO_ROOT/bench.py --mode chain --no-cpu-baseline --no-kernel-timers --no-extra --steps 10 --warmup 2 > $GRAFT_REPO_ROOT/$O/ch_$v.log 2>&1
  echo "== $v"; grep -E "jpeg_" $GRAFT_REPO_ROOT/$O/ch_${v}_kernel_stats.csv | cut -c1-150
done
rm -f $GRAFT_REPO_ROOT/$O/*_kernel_trace.csv $GRAFT_REPO_ROOT/$O/*agent_info.csv
