#!/usr/bin/env python
"""ATen / memcpy glue of one image-mode embed + detect step (32 x 768^2): torch-profiler table by call count, and the Python call sites of
every aten::copy_ / fill_ / cat that reaches the GPU.  usage: tools/prof_step.py   (GPU box)"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import videoseal_amd
from torch.profiler import profile, ProfilerActivity
model = videoseal_amd.build("videoseal_1.0", seed=0).eval().cuda()
model.chunk_size = 32
x = torch.rand(32, 3, 768, 768, device="cuda")
msgs = torch.randint(0, 2, (32, 256))
for _ in range(3):
    w = model.embed(x, msgs, is_video=False)["imgs_w"]; p = model.detect(w, is_video=True)["preds"]
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    w = model.embed(x, msgs, is_video=False)["imgs_w"]; p = model.detect(w, is_video=True)["preds"]
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=25, max_name_column_width=60))
sites = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::cat", "aten::zero_", "aten::index_put_", "aten::_to_copy", "aten::clone") and ev.device_time_total > 0:
        frames = [f for f in (ev.stack or []) if "videoseal_amd" in f or "bench.py" in f]
        sites[(ev.name, " <- ".join(s.strip().split("/")[-1] for s in frames[:3]))] += 1
for (name, where), n in sites.most_common(40):
    print(f"{n:4d} x {name:18s} {where}")
