#!/bin/bash
O=gpurun_out/r06p; mkdir -p $O
export VS_BENCH_COLLECTIVE=gloo HIP_VISIBLE_DEVICES=0
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 8 --steps 1 --warmup 1 --no-kernel-timers > $O/out.log 2> $O/err.log
echo rc=$?
grep -v "Gloo\|^$" $O/err.log | head -60
tail -c 600 $O/out.log
