#!/usr/bin/env python
"""Why the generator-step leg of the default bench run is slower than the same function in a fresh process.  The default run's sequence
(headline run with its CPU baseline, every extra leg at the default run's step counts), THEN the training leg for the first time in the process,
as is / after gc.collect() + torch.cuda.empty_cache().  usage: tools/diag_train_leg.py [--train-first]   (GPU box)"""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)


def show(tag):
    r = bench.gen_step_leg(dev)
    st = torch.cuda.memory_stats()
    print(f"{tag:44s} {r['ms_per_step']:7.2f} ms  host issue {r['host_issue_ms']:7.2f} ms   gc objects {len(gc.get_objects())}  "
          f"reserved {st['reserved_bytes.all.current'] >> 20} MiB  device mallocs so far {st['num_device_alloc']}  frees {st['num_device_free']}", flush=True)


if "--train-first" in sys.argv:
    show("fresh process")
bench.run(bench.parse_args([]))
legs = [("video", ["--mode", "video"], 5, 2), ("chain", ["--mode", "chain"], 5, 2), ("stream", ["--mode", "stream", "--frames", "1024"], 2, 1),
        ("chunkyseal", ["--card", "chunkyseal", "--size", "1024", "--batch", "16", "--detect-only"], 3, 1), ("pixelseal", ["--card", "pixelseal"], 5, 2),
        ("videoseal_0.0", ["--card", "videoseal_0.0"], 5, 2)]
for name, extra, k, w in legs:
    bench.run(bench.parse_args(["--steps", str(k), "--warmup", str(w), "--no-cpu-baseline", "--no-extra"] + extra))
show("after the default run's legs")
show("again")
gc.collect(); torch.cuda.empty_cache()
show("after gc.collect() + empty_cache()")
show("again")
