#!/bin/bash
# the K-slice epilogue with every load in flight at once: parity, then same-box A/B (vs_debug key via env VS_SPLITK_EPI=scalar) of the K-sliced legs
O=gpurun_out/r06ep; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "k_slice or k_slices or gemm1x1_pc or planes or patch" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for i in 1 2 3; do
  for F in vec scalar; do
    VS_SPLITK_EPI=$F python bench.py --mode chain --no-cpu-baseline --steps 30 --warmup 3 --no-kernel-timers --no-extra > $O/chain_${F}_$i.json 2>/dev/null
    VS_SPLITK_EPI=$F python bench.py --mode video --no-cpu-baseline --steps 30 --warmup 3 --no-kernel-timers --no-extra > $O/video_${F}_$i.json 2>/dev/null
  done
done
for F in vec scalar; do
  VS_SPLITK_EPI=$F python bench.py --no-cpu-baseline --steps 30 --warmup 3 --no-extra > $O/image_$F.json 2>/dev/null
  VS_SPLITK_EPI=$F python bench.py --mode stream --no-cpu-baseline --no-extra > $O/stream_$F.json 2>/dev/null
  VS_SPLITK_EPI=$F python bench.py --mode chain --no-cpu-baseline --steps 2 --warmup 1 --no-extra --dump-preds $O/p_$F.pt > /dev/null 2>&1
done
python - <<PY
import json,glob,torch
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"])
    except Exception as e: print(f, "unreadable", e)
try: print("chain preds identical:", torch.equal(torch.load("$O/p_vec.pt"), torch.load("$O/p_scalar.pt")))
except Exception as e: print("compare failed", e)
PY
rm -f $O/p_vec.pt $O/p_scalar.pt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O -o chain -- python $GRAFT_REPO_ROOT/bench.py --mode chain --no-cpu-baseline --no-kernel-timers --no-extra --steps 10 --warmup 2 > $GRAFT_REPO_ROOT/$O/chain.log 2>&1
rm -f $GRAFT_REPO_ROOT/$O/*_kernel_trace.csv $GRAFT_REPO_ROOT/$O/*agent_info.csv
grep -i "splitk" $GRAFT_REPO_ROOT/$O/chain_kernel_stats.csv | cut -c1-200
