#!/usr/bin/env python
"""Weight gradient of the U-Net's bottleneck 3x3 conv (384 -> 384 @32x32, 16 frames: vs_conv3x3_wgrad, matrix-core kernel on the implicit patch
matrix) and of the thin outer levels: time per call and TF-eq.  usage: tools/bench_wgrad.py   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoseal_amd import native as N
L, st = N.lib(), N.stream()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, H, W, ci, co, stride in ((16, 32, 32, 384, 384, 1), (16, 64, 64, 64, 64, 1), (16, 32, 32, 128, 128, 1), (16, 256, 256, 16, 16, 1), (16, 128, 128, 32, 32, 1)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B * H * W, ci, device="cuda", generator=g)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    dy = torch.randn(B * Ho * Wo, co, device="cuda", generator=g)
    part = torch.empty(int(L.vs_conv3x3_wgrad_partial_floats(co, ci, B, H, W, stride)), device="cuda")
    dw = torch.empty(co, 9 * ci, device="cuda")
    us = timed(lambda: N.check(L.vs_conv3x3_wgrad(N.ptr(dy), co, co, N.ptr(x), ci, B, H, W, stride, N.PAD_ZERO, N.ptr(part), N.ptr(dw), st), "wgrad"))
    fl = 2.0 * B * Ho * Wo * co * 9 * ci
    print(f"conv3x3 wgrad {ci}->{co} @{H}x{W} x{B}: {us:7.1f} us = {fl / us * 1e-6:6.1f} TF-eq (incl. the chunk reduction)")
