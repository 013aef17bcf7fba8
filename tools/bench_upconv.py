#!/usr/bin/env python
"""Upsample groups of the VideoSeal 1.0 U-Net at B = 32 (levels 0..2): fused one-kernel form vs cat2 + 9-tap GEMM + gather, HIP-event
times.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoseal_amd import native as N
from videoseal_amd.engine import Act, ConvW, HipEngine, pack_conv
import videoseal_amd
L = N.lib()
B = int(os.environ.get("B", 32))
model = videoseal_amd.build("videoseal_1.0").eval().cuda()
eng = model._engine()

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
    return best * 1e3

for (H, C1, C2, Co) in ((32, 384, 384, 64), (64, 64, 64, 32), (128, 32, 32, 16)):
    x = torch.randn(B, H, H, C1, device="cuda"); sk = torch.randn(B, H, H, C2, device="cuda")
    w = torch.randn(Co, C1 + C2, 3, 3, device="cuda") / (3 * (C1 + C2) ** 0.5)
    lw, lb = torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda")
    xa, sa = Act(x, B, H, H, C1, C1), Act(sk, B, H, H, C2, C2)
    lc = eng.new_act("b.lcat", B, H, H, C1 + C2)
    wz, cpz = pack_conv(w.permute(2, 3, 0, 1).reshape(9 * Co, C1 + C2)[:, :, None, None], lc.ld)
    cw = ConvW(wz, None, 9 * Co, 1, 1, cpz).with_split()
    z = eng.new_act("b.z", B, H, H, 9 * Co)
    out = eng.new_act("b.out", B, 2 * H, 2 * H, Co)
    st = N.stream()
    cat = lambda: N.check(L.vs_cat2_scale(N.ptr(xa.t), C1, xa.ld, N.ptr(sa.t), C2, sa.ld, 2 ** -0.5, lc.rows, N.ptr(lc.t), lc.ld, st), "cat")
    gemm = lambda: eng.conv(lc, cw, z)
    gath = lambda: N.check(L.vs_upconv_gather_ln(N.ptr(z.t), z.ld, B, H, H, Co, N.ptr(lw), N.ptr(lb), 1e-6, 1, N.ptr(out.t), out.ld, st), "g")
    t = [timeit(f) for f in (cat, gemm, gath)]
    line = f"{H}^2 {C1}+{C2}->{Co}: cat {t[0]:6.1f}  gemm9 {t[1]:6.1f}  gather+LN {t[2]:6.1f}  = {sum(t):6.1f} us"
    if L.vs_upconv_fused_supported(C1, C2, Co):
        fu = lambda: N.check(L.vs_upconv_fused(N.ptr(xa.t), C1, xa.ld, N.ptr(sa.t), C2, sa.ld, 2 ** -0.5, N.ptr(cw.split), B, H, H, Co,
                                               N.ptr(lw), N.ptr(lb), 1e-6, 1, N.ptr(out.t), out.ld, cw.arith, 16.0, 1.0 / (16.0 * cw.w_mul), st), "f")
        tf = timeit(fu)
        io = (x.numel() + sk.numel() + out.rows * Co) * 4
        line += f" | fused {tf:6.1f} us ({io / tf / 1e6:5.0f} GB/s of in+out)"
    print(line, flush=True)
