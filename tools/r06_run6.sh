#!/bin/bash
# round 6, GPU call 6: kernel statistics of the ChunkySeal detect leg (where do the 60 % outside the dominant GEMM go), stream leg with larger extractor batches
TAG=${1:-r06f}
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o chunky -- python $R/bench.py --detect-only --card chunkyseal --size 1024 --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers > $O/chunky_prof.log 2>&1
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv
cd $R
for DB in 32 64 128; do
  python bench.py --mode stream --frames 1024 --det-batch $DB --no-cpu-baseline --no-kernel-timers --steps 3 --warmup 1 > $O/stream_db$DB.json 2>/dev/null
  python bench.py --mode stream --frames 128 --det-batch $DB --no-cpu-baseline --no-kernel-timers --steps 5 --warmup 2 > $O/stream128_db$DB.json 2>/dev/null
done
for B in 32 64; do python bench.py --detect-only --batch $B --no-cpu-baseline --no-kernel-timers --steps 30 --warmup 3 > $O/detect_b$B.json 2>/dev/null; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], d["value"], d["ms_per_step"])
PY
