#!/bin/bash
# Round profile: bench lines, rocprofv3 kernel stats (image + video), three --pmc passes.  usage: tools/profile_round.sh <tag>   (GPU box)
TAG=${1:-r01x}
R=$PWD
export TMPDIR=/tmp VIDEOSEAL_TILE_CACHE=/tmp/tiles_$TAG.json
O=$R/gpurun_out/$TAG; mkdir -p $O
python bench.py --steps 30 --warmup 3 --no-extra > $O/bench_image.json 2> $O/bench_image.err
python bench.py --mode video --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_video.json 2> $O/bench_video.err
python bench.py --mode stream --no-cpu-baseline > $O/bench_stream.json 2> $O/bench_stream.err
python bench.py --mode chain --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_chain.json 2> $O/bench_chain.err
python bench.py --capi --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_capi.json 2> $O/bench_capi.err
python bench.py --detect-only --card chunkyseal --size 1024 --batch 16 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_chunkyseal.json 2> $O/bench_chunkyseal.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o img -- python $R/bench.py --no-extra --no-cpu-baseline --no-kernel-timers --steps 10 --warmup 2 > $O/img.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o chain -- python $R/bench.py --mode chain --no-cpu-baseline --no-kernel-timers --steps 10 --warmup 2 > $O/chain.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o vid -- python $R/bench.py --mode video --no-cpu-baseline --no-kernel-timers --steps 10 --warmup 2 > $O/vid.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stream -- python $R/bench.py --mode stream --no-cpu-baseline --no-kernel-timers --steps 2 --warmup 1 > $O/stream.log 2>&1
for P in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $P | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $P --output-format csv -d $O/pmc -o $N -- python $R/bench.py --no-extra --no-cpu-baseline --no-kernel-timers --steps 2 --warmup 1 > $O/pmc_$N.log 2>&1
done
for P in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $P --output-format csv -d $O/calib -o $P -- python $R/tools/calib_fetch.py > $O/calib_$P.log 2>&1
done
# round 6: the extractor alone and the ChunkySeal leg (kernel statistics), and the standing out-of-bounds probe -- the kernel / augmentation / end-to-end
# tests once more with every tensor its own hipMalloc (tests/conftest.py --no-caching-allocator; hipGraph tests skipped)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o detect -- python $R/bench.py --detect-only --no-cpu-baseline --no-kernel-timers --steps 20 --warmup 2 > $O/detect.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o chunky -- python $R/bench.py --detect-only --card chunkyseal --size 1024 --batch 16 --steps 12 --warmup 2 --no-cpu-baseline --no-kernel-timers > $O/chunky.log 2>&1
cd $R
python bench.py --detect-only --no-cpu-baseline --steps 40 --warmup 5 > $O/bench_detect_only.json 2> $O/bench_detect_only.err
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_aug.py tests/test_gpu_e2e.py tests/test_gpu_fwd.py -m gpu -q --no-caching-allocator > $O/pytest_no_caching_allocator.log 2>&1
tail -2 $O/pytest_no_caching_allocator.log
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv $O/pmc/*agent_info.csv $O/calib/*agent_info.csv
ls $O $O/pmc
