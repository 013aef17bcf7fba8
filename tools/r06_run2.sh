#!/bin/bash
# round 6, GPU call 2: box probe, full GPU suite on the new tree (red zones, in-group overlap, stated-size goldens, tight margins), out-of-bounds
# probe mode, detect / image A/B of the M0-neutral DMA against the round-5 library, ChunkySeal host RSS, 8-rank preflight if the host has room.
TAG=${1:-r06b}
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
(free -g; nproc; rocm-smi --showmeminfo vram 2>/dev/null | head -8) > $O/box.txt 2>&1
cat $O/box.txt | head -4
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_zdist.py::test_bench_eight_ranks_on_one_gpu_preflight -s > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|128-frame shard" $O/pytest_gpu.log | tail -4
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_aug.py tests/test_gpu_e2e.py tests/test_gpu_fwd.py -m gpu -q --no-caching-allocator > $O/pytest_oob_probe.log 2>&1
tail -2 $O/pytest_oob_probe.log
REF=$R/videoseal_amd/csrc/build_ab/libvideoseal_hip_ref.so
for i in 1 2; do
  python bench.py --no-cpu-baseline --detect-only --steps 30 --warmup 3 > $O/detect_new$i.json 2>/dev/null
  VIDEOSEAL_LIB=$REF python bench.py --no-cpu-baseline --detect-only --steps 30 --warmup 3 > $O/detect_ref$i.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-extra --steps 20 > $O/image_new$i.json 2>/dev/null
  VIDEOSEAL_LIB=$REF python bench.py --no-cpu-baseline --no-extra --steps 20 > $O/image_ref$i.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/detect_*.json")+glob.glob("$O/image_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], d["value"], d["ms_per_step"])
PY
/usr/bin/time -v python bench.py --detect-only --card chunkyseal --size 1024 --batch 16 --steps 3 --warmup 1 --no-cpu-baseline > $O/chunky.json 2> $O/chunky_time.txt
grep -E "Maximum resident|Elapsed" $O/chunky_time.txt
RSS_KB=$(grep "Maximum resident" $O/chunky_time.txt | awk '{print $NF}')
FREE_KB=$(awk '/MemAvailable/ {print $2}' /proc/meminfo)
echo "rss_kb=$RSS_KB free_kb=$FREE_KB"
if [ -n "$RSS_KB" ] && [ $((RSS_KB * 10)) -lt $FREE_KB ]; then
  VS_PREFLIGHT_OUT=$O/preflight8.json timeout 2400 python -m pytest tests/test_gpu_zdist.py -m gpu -q -s -k eight_ranks > $O/preflight8.log 2>&1
  grep -E "8-rank preflight|passed|failed" $O/preflight8.log | tail -3
else
  echo "8-rank preflight skipped: host memory"
fi
ls $O
