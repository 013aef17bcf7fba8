#!/bin/bash
# round 6, GPU call 1: the GPU suite with the decision log (measuring run: margins recorded, not judged), the stream leg at 128 / 1024 frames
# for every group size, kernel statistics of the generator training step.   usage: tools/r06_run1.sh <tag>   (GPU box)
TAG=${1:-r06a}
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
rm -f $O/decisions.jsonl
VS_DECISION_LOG=$O/decisions.jsonl VS_DECISION_DISCOVER=1 timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
for F in 128 1024; do
  for G in 1 2 4 8; do
    timeout 200 python bench.py --mode stream --frames $F --group $G --no-cpu-baseline --no-kernel-timers --steps 4 --warmup 2 > $O/stream_f${F}_g$G.json 2> $O/stream_f${F}_g$G.err
    python - <<PY
import json
try:
    d = json.loads(open("$O/stream_f${F}_g$G.json").read().strip().splitlines()[-1])
    print("stream frames=$F group=$G", d["value"], d["ms_per_step"])
except Exception as e:
    print("stream frames=$F group=$G failed", e)
PY
  done
done
timeout 200 python bench.py --mode stream --frames 128 --no-overlap --no-cpu-baseline --no-kernel-timers --steps 4 --warmup 2 > $O/stream_f128_serial.json 2>/dev/null
tail -c 400 $O/stream_f128_serial.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o train -- python $R/tools/bench_train.py > $O/train.log 2>&1
grep -m1 "^{'value'" $O/train.log
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv
ls $O
