#!/usr/bin/env python
"""Attention of the legacy videoseal_0.0 card's ViT extractor (vs_vit_attention: 16 x 16 tokens, 6 heads of 64; windows of 8 x 8 or global):
time per call of the kernel the library selects (matrix cores; VS_VIT_ATTN=valu in the environment = the vector kernel) and the detect
step of the card.  usage: tools/bench_vit.py [frames]   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import videoseal_amd
from videoseal_amd import native as N

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L, st = N.lib(), N.stream()
H = W = 16
heads, hd = 6, 64
D = heads * hd
g = torch.Generator().manual_seed(1)
qkv = torch.randn(B, H, W, 3 * D, generator=g).cuda()
out = torch.empty(B, H, W, D, device="cuda")


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


for win in (8, 0):
    T = win * win if win else H * W
    rel = torch.randn(2 * (win or H) - 1, hd, generator=g).cuda() * 0.3
    us = timed(lambda: N.check(L.vs_vit_attention(N.ptr(qkv), B, H, W, heads, hd, win, N.ptr(rel), N.ptr(rel), N.ptr(out), st), "attn"))
    flop = 4.0 * B * (H * W) * T * D          # q k^T and p v
    print(f"attention, {B} frames, {'window 8' if win else 'global'} ({T} keys per query): {us:.1f} us = {flop / us * 1e-6:.1f} TFLOP/s "
          f"[{os.environ.get('VS_VIT_ATTN', 'mfma')}]")
model = videoseal_amd.build("videoseal_0.0").eval().to("cuda")
frames = torch.rand(B, 3, 256, 256, device="cuda")
us = timed(lambda: model.detector(frames), 10)
print(f"videoseal_0.0 extractor forward (ViT, 12 blocks), {B} x 256x256: {us / 1e3:.2f} ms = {B / us * 1e6:.0f} frames/s")
