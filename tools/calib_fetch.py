#!/usr/bin/env python
"""Known-byte launches for calibrating the HBM counters (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): a device copy and a read-only
reduction of a 226.5 MB tensor (32 x 3 x 768 x 768 fp32: what one embed_tail launch reads), then the tail and resize kernels themselves.
The ratio known bytes / raw counter is the correction to apply to FETCH_SIZE for this access pattern (MI355X_MICROARCH.md's gfx950 note says x2
for 64-byte requests; 16-byte-per-lane row accesses may differ).  usage: tools/calib_fetch.py   (GPU box, under rocprofv3)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoseal_amd import native as N
from videoseal_amd.native import TailDesc
L = N.lib()
B, H, W, S = 32, 768, 768, 256
x = torch.rand(B, 3, H, W, device="cuda")
y = torch.empty_like(x)
torch.cuda.synchronize()
for _ in range(3):
    y.copy_(x)                       # elementwise copy kernel: 226.5 MB read + 226.5 MB written
    x.sum()                          # reduce kernel: 226.5 MB read
taps = (C.c_float * 43)(*([1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 2, 0, 2, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1] + [-1, 0, 1, -2, 0, 2, -1, 0, 1] + [1, 2, 1, 0, 0, 0, -1, -2, -1]))
delta = torch.randn(B, 1, S, S, device="cuda") * 0.1
out = torch.empty_like(x); pw = torch.empty(B, 1, H, W, device="cuda")
rgb = torch.empty(B, S, S, 4, device="cuda")
for att in (1, 0):
    d = TailDesc()
    d.imgs, d.out, d.preds_w, d.delta, d.hmap_lowres = N.ptr(x), N.ptr(out), N.ptr(pw), N.ptr(delta), None
    d.taps43 = C.cast(taps, C.c_void_p)
    d.F, d.H, d.W, d.S_h, d.S_w, d.Cd = B, H, W, S, S, 1
    d.step, d.video_mode, d.total_key = 1, 0, B
    d.attenuate, d.clamp, d.antialias = att, 1, 1
    d.scaling_i, d.scaling_w, d.io_u8 = 1.0, 0.2, 0
    for _ in range(3):
        N.check(L.vs_embed_tail(C.byref(d), N.stream()), "tail")
for _ in range(3):
    N.check(L.vs_resize_pre(N.ptr(x), B, 3, H, W, S, S, 1, N.ptr(rgb), 2.0, -1.0, None, 1, None, N.stream()), "r")
torch.cuda.synchronize()
print("known bytes per launch: copy 226.5 MB read + 226.5 MB written; sum 226.5 MB read; tail (JND / no JND) 226.5 MB read algorithmic, 226.5 + 75.5 MB written; "
      "resize 226.5 MB read, 33.5 MB written")
