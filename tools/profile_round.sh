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
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv $O/pmc/*agent_info.csv $O/calib/*agent_info.csv
ls $O $O/pmc
