#!/bin/bash
# round 6, final tree: full GPU suite, smoke, determinism soak, every bench mode + rocprofv3 + PMC (tools/profile_round.sh), the driver contract
# line, the 8-rank one-device preflight.   usage: tools/r06_final.sh <tag>
TAG=${1:-r06z}
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_zdist.py::test_bench_eight_ranks_on_one_gpu_preflight > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python tools/soak.py > $O/soak_determinism.log 2>&1; tail -3 $O/soak_determinism.log
tools/profile_round.sh $TAG > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("default line:", d["value"], d["ms_per_step"], d["roofline"]["frac"], {k[:24]:(v or {}).get("value") for k,v in d["configs"].items()})
PY
VS_PREFLIGHT_OUT=$O/preflight8.json timeout 2400 python -m pytest tests/test_gpu_zdist.py -m gpu -q -s -k eight_ranks > $O/preflight8.log 2>&1
grep -E "8-rank preflight|passed|failed" $O/preflight8.log | tail -2
