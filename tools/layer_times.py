#!/usr/bin/env python
"""Per-layer conv timing of one embed+detect step (HIP events), GPU box only."""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import videoseal_amd
B, S = int(os.environ.get("B", 32)), 768
model = videoseal_amd.build("videoseal_1.0").eval().cuda()
model.chunk_size = B
x = torch.rand(B, 3, S, S, device="cuda")
msgs = torch.randint(0, 2, (B, 256))
for _ in range(2):
    out = model.embed(x, msgs, is_video=False); model.detect(out["imgs_w"], is_video=True)
eng = model._engine(); eng.kernel_timers = []; eng.time_all_convs = True
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
out = model.embed(x, msgs, is_video=False); model.detect(out["imgs_w"], is_video=True)
e1.record(); torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, a, b, fl in eng.kernel_timers:
    t = agg.setdefault(name, [0, 0.0, 0.0]); t[0] += 1; t[1] += a.elapsed_time(b); t[2] += fl
tot = sum(v[1] for v in agg.values())
print(f"step {e0.elapsed_time(e1):.2f} ms, convs {tot:.2f} ms; tiles: {eng._tile_cache}")
for name, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:40s} x{n:3d} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF/s")
