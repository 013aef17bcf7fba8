#!/bin/bash
O=gpurun_out/r06pl; mkdir -p $O
PREV=$PWD/videoseal_amd/csrc/libvideoseal_prev.so
python bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-extra --dump-preds $O/p_new.pt > /dev/null 2>&1
VIDEOSEAL_LIB=$PREV python bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-extra --dump-preds $O/p_prev.pt > /dev/null 2>&1
python -c "
import torch; print('preds identical:', torch.equal(torch.load('$O/p_new.pt'), torch.load('$O/p_prev.pt')))"
rm -f $O/p_new.pt $O/p_prev.pt
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --detect-only --steps 40 --warmup 5 --no-kernel-timers --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', d['ms_per_step'])"
  VIDEOSEAL_LIB=$PREV python bench.py --no-cpu-baseline --detect-only --steps 40 --warmup 5 --no-kernel-timers --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prev (before grn_finish + pool_linear)', d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O -o det -- python $GRAFT_REPO_ROOT/bench.py --detect-only --no-cpu-baseline --no-kernel-timers --no-extra --steps 20 --warmup 2 > $GRAFT_REPO_ROOT/$O/det.log 2>&1
rm -f $GRAFT_REPO_ROOT/$O/*_kernel_trace.csv $GRAFT_REPO_ROOT/$O/*agent_info.csv
grep -iE "pool_linear|grn_finish" $GRAFT_REPO_ROOT/$O/det_kernel_stats.csv | cut -c1-160
cd $GRAFT_REPO_ROOT; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fwd.py -m gpu -q -x -k "pool or head or pixel_decoder or extractor" 2>&1 | tail -1
