#!/usr/bin/env python
"""The fused ConvNeXt block kernel (csrc/convnext_fused.hip) alone at the stage-0 / stage-1 shapes of 32 frames, statistics and apply launches;
VS_CNX_ABL=<bits> removes pieces (1 GELU, 2 pwconv2 MFMAs, 4 pwconv1 MFMAs, 8 output stores).  usage: tools/bench_cnx.py   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoseal_amd import native as N
from videoseal_amd.engine import pack_cnx_block
L = N.lib()
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
for C, HW in ((96, 4096), (192, 1024)):
    B = 32
    rows = B * HW
    g = torch.Generator().manual_seed(C)
    w1, b1 = torch.randn(4 * C, C, generator=g) * 0.1, torch.randn(4 * C, generator=g) * 0.1
    w2, beta = torch.randn(C, 4 * C, generator=g) * 0.05, torch.randn(4 * C, generator=g) * 0.1
    img, m1, m2 = pack_cnx_block(w1, b1, w2, beta, torch.device("cuda"))
    tn = (torch.randn(2, C // 16, rows, 16, generator=g) * 4).to(torch.float16).cuda()
    scale = (1 + torch.rand(B, 4 * C, generator=g)).cuda()
    b2 = torch.zeros(C, device="cuda")
    cur = torch.randn(rows, C, device="cuda")
    part = torch.empty(rows // 32, 4 * C, device="cuda")
    am1, am2 = 1 / (16 * m1), 1 / m2
    st = lambda: N.check(L.vs_cnx_block(N.ptr(tn), N.ptr(img), C, rows, HW, 1, am1, am2, None, 0, None, None, 0, None, 0, N.ptr(part), N.stream()), "s")
    ap = lambda: N.check(L.vs_cnx_block(N.ptr(tn), N.ptr(img), C, rows, HW, 0, am1, am2, N.ptr(scale), 4 * C, N.ptr(b2), N.ptr(cur), C, N.ptr(cur), C, None, N.stream()), "a")
    flop1 = 2 * rows * C * 4 * C
    sts = lambda: N.check(L.vs_cnx_block(N.ptr(tn), N.ptr(img), C, rows, HW, 3, am1, am2, None, 0, None, None, 0, None, 0, N.ptr(part), N.stream()), "s")
    aps = lambda: N.check(L.vs_cnx_block(N.ptr(tn), N.ptr(img), C, rows, HW, 2, am1, am2, N.ptr(scale), 4 * C, N.ptr(b2), N.ptr(cur), C, N.ptr(cur), C, None, N.stream()), "a")
    # ablation bits only act in a build with -DVS_CNX_ABLATION (make EXTRA=-DVS_CNX_ABLATION; VIDEOSEAL_LIB=<that build>)
    for abl in ([0] if "--abl" not in sys.argv else [0, 1, 2, 4, 6, 7, 8, 16, 17, 23]):
        os.environ["VS_CNX_ABL"] = str(abl)
        ts, ta, tss, tas = timeit(st), timeit(ap), timeit(sts), timeit(aps)
        print(f"C={C} rows={rows} abl={abl:2d}: pipelined stats {ts*1e3:6.1f} us ({flop1/ts/1e9:6.0f} TF-eq) apply {ta*1e3:6.1f} us ({2*flop1/ta/1e9:6.0f} TF-eq)"
              f"   serial stats {tss*1e3:6.1f} us apply {tas*1e3:6.1f} us", flush=True)
