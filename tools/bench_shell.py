#!/usr/bin/env python
"""Shell kernels alone: resize_pre (fp32 NCHW / uint8 RGB24, antialias on/off) and embed_tail variants, GB/s of algorithmic bytes."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoseal_amd import native as N
L = N.lib()
B, H, W, S = 32, 768, 768, 256
x = torch.rand(B, 3, H, W, device="cuda") * 0.5 + torch.nn.functional.interpolate(torch.rand(B, 3, H // 16, W // 16, device="cuda"), size=(H, W), mode="bilinear") * 0.5
xu = (x * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
rgb = torch.empty(B, S, S, 4, device="cuda"); key = torch.empty(B, S, S, 4, device="cuda")
ymat = (C.c_float * 3)(0.299, 0.587, 0.114)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best
for aa in (1, 0):
    ms = timeit(lambda: N.check(L.vs_resize_pre(N.ptr(x), B, 3, H, W, S, S, aa, N.ptr(rgb), 2.0, -1.0, None, 1, None, N.stream()), "r"))
    print(f"resize_pre fp32 aa={aa}: {ms*1e3:7.1f} us  {(x.numel()*4 + rgb.numel()*4)/ms/1e6:7.0f} GB/s   (row-streaming separable form)")
    L.vs_debug_set(0, 1)
    ms = timeit(lambda: N.check(L.vs_resize_pre(N.ptr(x), B, 3, H, W, S, S, aa, N.ptr(rgb), 2.0, -1.0, None, 1, None, N.stream()), "r"))
    L.vs_debug_set(0, 0)
    print(f"resize_pre fp32 aa={aa}: {ms*1e3:7.1f} us  {(x.numel()*4 + rgb.numel()*4)/ms/1e6:7.0f} GB/s   (32 x 8 tile form)")
    ms = timeit(lambda: N.check(L.vs_resize_pre_u8(N.ptr(xu), B, H, W, S, S, aa, N.ptr(rgb), 2.0, -1.0, None, 1, None, N.stream()), "r"))
    print(f"resize_pre u8   aa={aa}: {ms*1e3:7.1f} us  {(xu.numel() + rgb.numel()*4)/ms/1e6:7.0f} GB/s")
y = torch.empty_like(x)
ms = timeit(lambda: y.copy_(x))
print(f"torch copy 302 MB -> 302 MB: {ms*1e3:7.1f} us {2*x.numel()*4/ms/1e6:7.0f} GB/s (HBM copy reference)")
ms = timeit(lambda: x.sum())
print(f"torch sum 302 MB: {ms*1e3:7.1f} us {x.numel()*4/ms/1e6:7.0f} GB/s (HBM read reference)")

# ---- embed tail
import math
from videoseal_amd.native import TailDesc
taps = (C.c_float * 43)(*([1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 2, 0, 2, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1] + [-1, 0, 1, -2, 0, 2, -1, 0, 1] + [1, 2, 1, 0, 0, 0, -1, -2, -1]))   # jnd.py:24-41
delta = torch.randn(B, 1, S, S, device="cuda") * 0.1
hm = torch.rand(B, S, S, device="cuda")
out = torch.empty_like(x); outu = torch.empty_like(xu); pw = torch.empty(B, 1, H, W, device="cuda")
def tail(u8, att, low, with_pw, aa=1, step=1, variant=0):
    d = TailDesc()
    d.variant = variant
    src, dst = (xu, outu) if u8 else (x, out)
    d.imgs, d.out, d.preds_w = N.ptr(src), N.ptr(dst), (N.ptr(pw) if with_pw else None)
    d.delta, d.hmap_lowres = N.ptr(delta), (N.ptr(hm) if low else None)
    d.taps43 = C.cast(taps, C.c_void_p)
    d.F, d.H, d.W, d.S_h, d.S_w, d.Cd = B, H, W, S, S, 1
    d.step, d.video_mode, d.total_key = step, 0, B // step
    d.attenuate, d.clamp, d.antialias = att, 1, aa
    d.scaling_i, d.scaling_w, d.io_u8 = 1.0, 0.2, int(u8)
    return lambda: N.check(L.vs_embed_tail(C.byref(d), N.stream()), "tail")
for name, u8, att, low, wpw in [("fp32 full JND + preds_w", 0, 1, 0, 1), ("fp32 full JND", 0, 1, 0, 0), ("fp32 lowres JND", 0, 1, 1, 0), ("fp32 no JND", 0, 0, 0, 0),
                                ("u8 full JND", 1, 1, 0, 0), ("u8 lowres JND", 1, 1, 1, 0), ("u8 no JND", 1, 0, 0, 0)]:
    ms = timeit(tail(u8, att, low, wpw))
    nbytes = (2 * xu.numel() if u8 else 2 * x.numel() * 4) + (pw.numel() * 4 if wpw else 0)
    print(f"embed_tail {name:24s}: {ms*1e3:7.1f} us  {nbytes/ms/1e6:7.0f} GB/s")
# the forms of the fp32 tail side by side (vs_tail_desc_t::variant): 1 = 43-tap JND on 16-row tiles, 2 = separable stencils on 16-row tiles,
# 4 = row-streaming strips (default); VS_TAIL_STRIP=<rows> overrides the strip height of the streaming form
for name, att, low in [("fp32 full JND", 1, 0), ("fp32 lowres JND", 1, 1), ("fp32 no JND", 0, 0)]:
    for variant in (1, 2, 4):
        ms = timeit(tail(0, att, low, 0, variant=variant))
        print(f"embed_tail {name:18s} variant {variant}: {ms*1e3:7.1f} us  {2 * x.numel() * 4/ms/1e6:7.0f} GB/s")
