#!/bin/bash
# round 6, last call: the 8-rank preflight on the final bench.py, a same-box A/B of the streaming extractor batch, the driver contract line
TAG=${1:-r06zz}
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
VS_PREFLIGHT_OUT=$O/preflight8.json timeout 2400 python -m pytest tests/test_gpu_zdist.py -m gpu -q -s -k "eight_ranks or two_ranks" > $O/preflight8.log 2>&1
grep -E "8-rank preflight|passed|failed" $O/preflight8.log | tail -3
for i in 1 2; do
  for DB in 32 128; do
    python bench.py --mode stream --frames 1024 --det-batch $DB --no-cpu-baseline --no-kernel-timers --steps 3 --warmup 1 > $O/stream_db${DB}_$i.json 2>/dev/null
    python bench.py --mode stream --frames 128 --det-batch $DB --no-cpu-baseline --no-kernel-timers --steps 5 --warmup 2 > $O/stream128_db${DB}_$i.json 2>/dev/null
  done
done
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/stream*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"])
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("default line:", d["value"], d["ms_per_step"], d["roofline"]["frac"], {k[:24]:(v or {}).get("value") for k,v in d["configs"].items()})
PY
