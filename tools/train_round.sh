#!/bin/bash
# Training-path measurement of a round: backward / training test files, the world-2 DDP test, the generator step with its torch-profiler table,
# the same step with the patch-matrix weight gradients (A/B), the detector-only step.  usage: tools/train_round.sh <tag>   (GPU box)
TAG=${1:-r03t}
R=$PWD
O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_bwd_unet.py tests/test_gpu_train.py -x -q > $O/tests_bwd.log 2>&1
tail -4 $O/tests_bwd.log
timeout 600 python -m pytest tests/test_gpu_zdist.py -x -q -k "two_ranks_on_one_gpu or ddp_wrapped" > $O/tests_ddp.log 2>&1
tail -4 $O/tests_ddp.log
timeout 300 python tools/bench_train.py --torch-profile > $O/train_profile.log 2>&1
grep -m1 "^{'value'" $O/train_profile.log
VIDEOSEAL_DIRECT_WGRAD=0 timeout 300 python tools/bench_train.py > $O/train_patch_matrix_wgrad.log 2>&1
grep -m1 "^{'value'" $O/train_patch_matrix_wgrad.log
timeout 300 python tools/bench_bwd.py 16 > $O/detector_step.log 2>&1
tail -1 $O/detector_step.log
