#!/usr/bin/env python
"""embed tail: every form x strip height x frame count on ONE box (the boxes of the pool differ by more than the forms do).
`python tools/bench_tail_sweep.py`  ->  one line per (frames, case): 43-tap tiles (v1), separable tiles (v2), row-streaming strips (s<rows>)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoseal_amd import native as N
from videoseal_amd.native import TailDesc

L = N.lib()
H = W = 768
S = 256
taps = (C.c_float * 43)(*([1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 2, 0, 2, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1] + [-1, 0, 1, -2, 0, 2, -1, 0, 1] + [1, 2, 1, 0, 0, 0, -1, -2, -1]))


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


for F_ in (16, 32, 128):
    x = torch.rand(F_, 3, H, W, device="cuda")
    out = torch.empty_like(x)
    delta = torch.randn(F_, 1, S, S, device="cuda") * 0.1
    hm = torch.rand(F_, S, S, device="cuda")
    pw = torch.empty(F_, 1, H, W, device="cuda")
    for name, att, low, wpw in (("full JND", 1, 0, 0), ("full JND + preds_w", 1, 0, 1), ("lowres JND", 1, 1, 0)):
        def mk(variant):
            d = TailDesc()
            d.imgs, d.out, d.preds_w = N.ptr(x), N.ptr(out), (N.ptr(pw) if wpw else None)
            d.delta, d.hmap_lowres, d.taps43 = N.ptr(delta), (N.ptr(hm) if low else None), C.cast(taps, C.c_void_p)
            d.F, d.H, d.W, d.S_h, d.S_w, d.Cd = F_, H, W, S, S, 1
            d.step, d.video_mode, d.total_key = 1, 0, F_
            d.attenuate, d.clamp, d.antialias = att, 1, 1
            d.scaling_i, d.scaling_w, d.io_u8, d.variant = 1.0, 0.2, 0, variant
            return lambda: N.check(L.vs_embed_tail(C.byref(d), N.stream()), "tail")
        nbytes = 2 * x.numel() * 4 + (pw.numel() * 4 if wpw else 0)
        row = [f"F={F_:3d} {name:20s}"]
        for v in (1, 2):
            ms = timeit(mk(v))
            row.append(f"v{v} {ms*1e3:6.1f}us")
        for strip in (32, 48, 64, 96, 128):
            L.vs_debug_set(2, strip)
            ms = timeit(mk(4))
            row.append(f"s{strip} {ms*1e3:6.1f}us")
        L.vs_debug_set(2, 0)
        ms = timeit(mk(0))
        row.append(f"default {ms*1e3:6.1f}us {nbytes/ms/1e6:5.0f}GB/s")
        print("  ".join(row), flush=True)
