#!/usr/bin/env python
"""Runs only the bottleneck conv (f32 and split back-ends) a few times: target for rocprofv3 --pmc passes."""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videoseal_amd import native as N
from videoseal_amd.engine import Act, ConvW, pack_conv
from tools.bench_conv import Eng
eng = Eng()
B, Cin, H, W, Cout, k = 32, 384, 32, 32, 384, 3
g = torch.Generator().manual_seed(1)
x = torch.randn(B, H, W, Cin, generator=g).cuda()
w = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).cuda()
xa = Act(x, B, H, W, Cin, Cin)
wt, cp = pack_conv(w, Cin)
cw = ConvW(wt, None, Cout, k, k, cp).with_split()
out = eng.new_act("o", B, H, W, Cout)
for hint in (N.CONV_FORCE_F32 | 1, N.CONV_FORCE_SPLIT | 1):
    eng.use_split = not (hint & N.CONV_FORCE_F32)
    for _ in range(3):
        eng.conv(xa, cw, out, pad=1, tile_hint=hint)
torch.cuda.synchronize()
