#!/usr/bin/env python
"""dwconv7 + LayerNorm of the four ConvNeXt stages (B = 32 frames at 256^2 proc): time and effective GB/s (in + out).
VS_DWCONV=<n> picks the kernel (unset: the library's rule; 0: one-row kernel; 1..4: LDS-tiled configurations) -- one process per value."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoseal_amd import native as N
L = N.lib()
B = 32
print("VS_DWCONV =", os.environ.get("VS_DWCONV", "auto"))
SHAPES = ((64, 96), (32, 192), (16, 384), (8, 768))
if "b16" in sys.argv:          # the chain leg's extractor pass
    B = 16
if "chunky" in sys.argv:        # ChunkySeal's extractor, 16 frames: channel strides padded to 32 (engine._xld)
    B = 16
    SHAPES = ((127, 384), (63, 736), (31, 1472), (15, 2912))
for HW, Cc in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(HW)
    x = torch.randn(B, HW, HW, Cc, device="cuda", generator=g); out = torch.empty_like(x)
    wdw = torch.randn(49, Cc, device="cuda", generator=g); v = [torch.randn(Cc, device="cuda", generator=g) for _ in range(3)]
    fn = lambda: L.vs_dwconv7_ln(N.ptr(x), B, HW, HW, Cc, Cc, N.ptr(wdw), N.ptr(v[0]), N.ptr(v[1]), N.ptr(v[2]), 1e-6, N.ptr(out), Cc, N.stream())
    if fn() != 0:
        print(f'dwconv7_ln {HW}x{HW} C={Cc}: unsupported by this configuration'); continue
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    bits = int(out.view(torch.int32).to(torch.int64).sum())          # every configuration must print the same number (bit-identical outputs)
    print(f"dwconv7_ln {HW}x{HW} C={Cc}: {best*1e3:7.1f} us  {2*x.numel()*4/best/1e6:6.0f} GB/s  ({49*x.numel()*2/best/1e9:5.1f} TFLOP/s fp32 FMA)  bits {bits}")
