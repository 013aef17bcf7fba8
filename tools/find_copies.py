#!/usr/bin/env python
"""Which torch-level ops of one embed + detect step end in a copy / fill / reduce launch (the launches that are not ours).
`python tools/find_copies.py [--mode image|video]`  ->  aten ops with device time, and every Memcpy / Memset with its size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import videoseal_amd
from bench import synthetic_batch

mode = sys.argv[sys.argv.index("--mode") + 1] if "--mode" in sys.argv else "image"
dev = torch.device("cuda:0")
model = videoseal_amd.build("videoseal_1.0", seed=0).eval().to(dev)
frames = synthetic_batch(32, 768, dev, seed=1000)
vid = mode == "video"
msgs = torch.randint(0, 2, (1 if vid else 32, model.embedder.cfg.nbits), generator=torch.Generator().manual_seed(5))
model.chunk_size = 32


def step():
    w = model.embed(frames, msgs, is_video=vid)["imgs_w"]
    return model.detect(w, is_video=vid)["preds"]


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
print("aten ops that launched device work (one step):")
for e in sorted(ka, key=lambda e: -e.device_time_total):
    if e.key.startswith("aten::") and e.device_time_total > 0:
        print(f"  {e.key:34s} x{e.count:3d}  dev {e.device_time_total:8.1f} us  shapes {str(e.input_shapes)[:90]}")
print("device-side copies / fills / torch kernels:")
for e in prof.events():
    n = e.name
    if e.device_type == torch.autograd.DeviceType.CUDA and ("Memcpy" in n or "Memset" in n or "copyBuffer" in n or "fillBuffer" in n or n.startswith("void at::")):
        print(f"  {n[:70]:70s} {e.device_time:7.1f} us")
# python stacks of the copy ops
seen = {}
for e in prof.events():
    if e.name in ("aten::copy_", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::sum", "aten::any", "aten::all", "aten::isfinite", "aten::cat", "aten::clone") and e.stack:
        key = (e.name, tuple(s for s in e.stack if "videoseal_amd" in s or "bench.py" in s)[:3])
        seen[key] = seen.get(key, 0) + 1
print("python call sites:")
for (name, st), c in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"  x{c:3d} {name:16s} {' <- '.join(s.strip()[-80:] for s in st)}")
