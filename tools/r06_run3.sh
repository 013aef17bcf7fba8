#!/bin/bash
# round 6, GPU call 3: GRN finish folded into pwconv2 (tests + same-box A/B by VIDEOSEAL_GRN_FOLD), vectorised pool_linear, 8-rank preflight
TAG=${1:-r06c}
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "grn or gemm1x1_pc or pool_linear or gemm_planes" > $O/pytest_kernels.log 2>&1
tail -3 $O/pytest_kernels.log
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fwd.py tests/test_gpu_shim.py -m gpu -q -x > $O/pytest_e2e.log 2>&1
tail -3 $O/pytest_e2e.log
REF=$R/videoseal_amd/csrc/build_ab/libvideoseal_hip_ref.so
for i in 1 2 3; do
  VIDEOSEAL_GRN_FOLD=1 python bench.py --no-cpu-baseline --detect-only --steps 40 --warmup 5 --no-kernel-timers > $O/detect_fold$i.json 2>/dev/null
  VIDEOSEAL_GRN_FOLD=0 python bench.py --no-cpu-baseline --detect-only --steps 40 --warmup 5 --no-kernel-timers > $O/detect_nofold$i.json 2>/dev/null
done
VIDEOSEAL_LIB=$REF python bench.py --no-cpu-baseline --detect-only --steps 40 --warmup 5 --no-kernel-timers > $O/detect_r05lib.json 2>/dev/null
python bench.py --no-cpu-baseline --capi --steps 20 --warmup 3 > $O/capi_new.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/detect_*.json")+glob.glob("$O/capi_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], d["value"], d["ms_per_step"])
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o detect -- python $R/bench.py --detect-only --no-cpu-baseline --no-kernel-timers --steps 20 --warmup 2 > $O/detect_prof.log 2>&1
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv
cd $R
VS_PREFLIGHT_OUT=$O/preflight8.json timeout 2400 python -m pytest tests/test_gpu_zdist.py -m gpu -q -s -k eight_ranks > $O/preflight8.log 2>&1
grep -E "8-rank preflight|passed|failed|Error|error" $O/preflight8.log | tail -5
ls $O
