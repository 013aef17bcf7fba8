#!/bin/bash
# Profile of the detector fine-tuning step (videoseal_amd.training.DetectorStep): time per step with both weight-gradient kernels, then
# rocprofv3 kernel stats of the step.  usage: tools/profile_bwd.sh <tag> [batch]   (GPU box)
TAG=${1:-r03a}
B=${2:-16}
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
python tools/bench_bwd.py $B > $O/bwd_fma.log 2>&1
VS_WGRAD=mfma timeout 120 python -m pytest tests/test_gpu_bwd.py -q > $O/bwd_mfma_tests.log 2>&1
VS_WGRAD=mfma python tools/bench_bwd.py $B > $O/bwd_mfma.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o bwd -- python $R/tools/bench_bwd.py $B > $O/bwd_prof.log 2>&1
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv
tail -2 $O/bwd_fma.log $O/bwd_mfma_tests.log $O/bwd_mfma.log
