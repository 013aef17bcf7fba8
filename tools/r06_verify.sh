#!/bin/bash
# the two commands the driver runs at round end, on the final commit
O=gpurun_out/r06v; mkdir -p $O
timeout 1700 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("default line:", d["value"], d["ms_per_step"], d["roofline"]["frac"], {k[:24]:(v or {}).get("value") for k,v in d["configs"].items()})
PY
