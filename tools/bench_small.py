#!/usr/bin/env python
"""Thin full-resolution 3x3 layers of the U-Net: patch kernel (tile 10) vs the persistent thin-layer kernel (tile 20)."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoseal_amd import native as N
from videoseal_amd.engine import Act, ConvW, pack_conv
from tools.bench_conv import Eng

TILES = (10, 0x44, 0x144, 0x244, 0x444, 0x344, 0x744)

def run(name, B, C, H, W, Co, two, reps=30):
    eng = Eng()
    g = torch.Generator().manual_seed(1)
    ld = max(4, (C + 3) // 4 * 4)
    x = torch.zeros(B, H, W, ld); x[..., :C] = torch.randn(B, H, W, C, generator=g); x = x.cuda()
    x2 = torch.randn(B, H, W, 16, generator=g).cuda()
    w = (torch.randn(Co, C, 3, 3, generator=g) / math.sqrt(C * 9)).cuda(); w2 = (torch.randn(Co, 16, 1, 1, generator=g) / 4).cuda()
    xa, xa2 = Act(x, B, H, W, C, ld), Act(x2, B, H, W, 16, 16)
    wt, cp = pack_conv(w, ld); wt2, cp2 = pack_conv(w2, 16)
    cw, cw2 = ConvW(wt, torch.zeros(Co).cuda(), Co, 3, 3, cp), ConvW(wt2, torch.zeros(Co).cuda(), Co, 1, 1, cp2)
    out = eng.new_act("o", B, H, W, Co)
    res = {}
    for rnd in range(4):
        for t in TILES:
            kw = dict(pad=1, act=N.ACT_RELU, tile_hint=t)
            if two: kw.update(in2=xa2, w2=cw2)
            for _ in range(2): eng.conv(xa, cw, out, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): eng.conv(xa, cw, out, **kw)
            e1.record(); torch.cuda.synchronize()
            res[t] = min(res.get(t, 1e9), e0.elapsed_time(e1) / reps)
    gb = (x.numel() + out.t.numel()) * 4 / 1e9
    print(f"{name:28s} " + "  ".join(f"[{t:#x}] {v*1e3:6.1f}us" for t, v in res.items()) + f"   ({gb/res[0x44]*1e3:5.0f} GB/s of in+out)", flush=True)

if __name__ == "__main__":
    run("16->16 @256^2 B32", 32, 16, 256, 256, 16, False)
    run("16->16 @256^2 B32 +1x1", 32, 16, 256, 256, 16, True)
    run("1->16 @256^2 B32", 32, 1, 256, 256, 16, False)
    run("16->32 @128^2 B32", 32, 16, 128, 128, 32, False)
    run("16->16 @256^2 B8", 8, 16, 256, 256, 16, False)
