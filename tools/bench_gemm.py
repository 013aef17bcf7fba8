#!/usr/bin/env python
"""1x1 GEMMs of the ConvNeXt extractor (B=32 frames at 256x256): generic kernel tiles vs the wave-specialised GEMM (17/18), with K split."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoseal_amd import native as N
from videoseal_amd.engine import Act, ConvW, pack_conv
from tools.bench_conv import Eng

HI = N.CONV_TILE_HI

def run(name, B, HW, K, Nn, variants, grn=False, reps=30, planes=False):
    """planes: tile codes 24 / 25 (HI | 8, HI | 9) read the activations as pre-split planes; with grn the conversion pass that applies the
    GRN scale (vs_to_planes_affine) is timed separately and printed next to them"""
    eng = Eng()
    eng.arith = 2
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B * HW * HW, K, generator=g).cuda()
    w = (torch.randn(Nn, K, 1, 1, generator=g) / math.sqrt(K)).cuda()
    xa = Act(x, B, HW, HW, K, K)
    wt, cp = pack_conv(w, K)
    cw = ConvW(wt, torch.zeros(Nn).cuda(), Nn, 1, 1, cp)
    out = eng.new_act("o", B, HW, HW, Nn)
    kw = {}
    if grn:
        kw = dict(a_scale=(1 + 0.1 * torch.randn(B, K, generator=g)).cuda(), a_scale_ld=K, a_shift=torch.zeros(K).cuda(), res=out)
    flops = 2.0 * B * HW * HW * K * Nn
    best = {v: 1e9 for v in variants}
    pl, conv_us = None, 0.0
    if planes:
        pl = eng.buf("pl", xa.rows * K).view(torch.int16)
        sc = kw.get("a_scale")
        cvt = lambda: N.check(eng.lib.vs_to_planes_affine(N.ptr(xa.t), xa.rows, K, K, 16.0, N.ptr(sc) if grn else None, K,
                                                          N.ptr(kw["a_shift"]) if grn else None, HW * HW, N.ptr(pl), N.stream()), "cvt")
        for _ in range(3): cvt()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): cvt()
        e1.record(); torch.cuda.synchronize()
        conv_us = e0.elapsed_time(e1) / reps * 1e3
    for rnd in range(4):
        for v in variants:
            t, sk = v
            kv = dict(kw)
            if (t & 0x4f) in (0x48, 0x49):
                kv = dict(res=kw.get("res"), in_pl=pl)
            for _ in range(2): eng.conv(xa, cw, out, tile_hint=t, split_k=sk, act=N.ACT_GELU if not grn else 0, arith=2, **kv)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): eng.conv(xa, cw, out, tile_hint=t, split_k=sk, act=N.ACT_GELU if not grn else 0, arith=2, **kv)
            e1.record(); torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) / reps)
    if planes:
        name += f" [to_planes {conv_us:5.1f}us]"
    print(name + ": " + "  ".join(f"[t{(t & 15) + (16 if t & HI else 0)}/{t >> 8:x} sk{sk}] {ms*1e3:6.1f}us {flops/ms/1e9:5.0f}TF" for (t, sk), ms in best.items()), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "abl":
        for t in (HI | 1, HI | 2):
            V = [(5, 1), (1, 1), (t, 1)]
            run("s2 pw1 384->1536 M=8192 ", 32, 16, 384, 1536, V)
            run("chunky s2 1472->5888 M=15376", 16, 31, 1472, 5888, V, reps=5)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ksweep":      # fixed cost per launch of the wave-specialised GEMM: time vs K at M = 8192, N = 1536
        for K in (64, 128, 256, 384, 768, 1536):
            run(f"M=8192 N=1536 K={K:4d}", 32, 16, K, 1536, [(HI | 2, 1), (HI | 1, 1), (1, 1)])
        for K in (64, 384):
            run(f"M=8192 N=1536 K={K:4d} ablations (none / no act / no stores / neither)", 32, 16, K, 1536,
                [(HI | 2, 1), (HI | 2 | 0x4000, 1), (HI | 2 | 0x2000, 1), (HI | 2 | 0x6000, 1)])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ksweep2":     # round 5: the same for the shapes of stage 2 as they run today (pwconv2 with the GRN transform + residual; pwconv1 on planes)
        for K in (128, 384, 768, 1536, 3072):
            run(f"pw2-like M=8192 N=384 K={K:4d} GRN+res", 32, 16, K, 384, [(HI | 2, 1), (HI | 2, 2), (HI | 10, 1), (HI | 1, 1)], grn=True)
        for K in (128, 384, 768, 1536):
            run(f"pw1-like M=8192 N=1536 K={K:4d} planes", 32, 16, K, 1536, [(HI | 8, 1), (HI | 2, 1)], planes=True)
        for K in (384,):        # needs the library built with EXTRA=-DVS_KERNEL_ABLATION (VIDEOSEAL_LIB): none / no activation / no output stores / neither
            run(f"pw1-like K={K} ablations planes", 32, 16, K, 1536, [(HI | 8, 1), (HI | 8 | 0x4000, 1), (HI | 8 | 0x2000, 1), (HI | 8 | 0x6000, 1)], planes=True)
            run(f"pw2-like K=1536 ablations", 32, 16, 1536, 384, [(HI | 2, 2), (HI | 2 | 0x2000, 2), (HI | 10, 1), (HI | 10 | 0x2000, 1)], grn=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "bigm":        # the wave-specialised GEMM's tiles at the row counts of the stream / detect-batch-256 legs and of the image step
        run("s2 pw2 1536->384 M=65536", 256, 16, 1536, 384, [(HI | 2, 1), (HI | 10, 1), (HI | 1, 1)], grn=True, reps=10)
        run("s3 pw2 3072->768 M=16384", 256, 8, 3072, 768, [(HI | 2, 1), (HI | 10, 1), (HI | 2, 2), (HI | 1, 1)], grn=True, reps=10)
        run("s3 pw1 768->3072 M=16384", 256, 8, 768, 3072, [(HI | 2, 1), (HI | 10, 1), (HI | 1, 1)], reps=10)
        run("s3 pw1 768->3072 M=2048 ", 32, 8, 768, 3072, [(HI | 2, 1), (HI | 10, 1), (HI | 1, 1)])
        run("s3 pw2 3072->768 M=2048 ", 32, 8, 3072, 768, [(HI | 2, 4), (HI | 10, 1), (HI | 10, 2), (HI | 10, 4)], grn=True)
        run("up  gemm 1152->768 M=8192", 32, 16, 1152, 768, [(HI | 2, 1), (HI | 10, 1), (HI | 1, 1)])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "b16":        # the extractor's GEMMs at 16 frames (the chain leg, configs[2]): K-slice count / tile alternatives
        run("s2 pw1 384->1536 M=4096 ", 16, 16, 384, 1536, [(HI | 2, 1), (HI | 10, 1), (HI | 1, 1)])
        run("s2 pw2 1536->384 M=4096 ", 16, 16, 1536, 384, [(HI | 2, 4), (HI | 2, 2), (HI | 10, 2), (HI | 10, 1), (HI | 10, 4), (HI | 1, 4), (HI | 1, 2)], grn=True)
        run("s3 pw1 768->3072 M=1024 ", 16, 8, 768, 3072, [(HI | 2, 1), (HI | 10, 1), (HI | 1, 1)])
        run("s3 pw2 3072->768 M=1024 ", 16, 8, 3072, 768, [(HI | 2, 8), (HI | 2, 4), (HI | 10, 4), (HI | 10, 8), (HI | 1, 8), (HI | 1, 4)], grn=True)
        run("down 768->384 M=4096    ", 16, 16, 768, 384, [(HI | 2, 4), (HI | 2, 2), (HI | 10, 2), (HI | 10, 1), (HI | 1, 2)])
        run("down 1536->768 M=1024   ", 16, 8, 1536, 768, [(HI | 2, 8), (HI | 2, 4), (HI | 10, 4), (HI | 10, 2), (HI | 1, 4)])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "b32":        # the same at 32 frames: the down-samplers (dense GEMMs on the LayerNorm's patch matrix) and stage 3
        run("down 384->192 M=32768   ", 32, 32, 384, 192, [(HI | 2, 1), (HI | 10, 1), (HI | 1, 1)])
        run("down 768->384 M=8192    ", 32, 16, 768, 384, [(HI | 2, 2), (HI | 10, 1), (HI | 1, 2), (HI | 1, 1)])
        run("down 1536->768 M=2048   ", 32, 8, 1536, 768, [(HI | 2, 4), (HI | 10, 2), (HI | 10, 4), (HI | 10, 1), (HI | 1, 4), (HI | 1, 2)])
        run("s3 pw1 768->3072 M=2048 ", 32, 8, 768, 3072, [(HI | 2, 1), (HI | 10, 1), (HI | 1, 1)])
        run("s3 pw2 3072->768 M=2048 ", 32, 8, 3072, 768, [(HI | 2, 4), (HI | 2, 2), (HI | 10, 2), (HI | 10, 4), (HI | 1, 4)], grn=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "planes":      # all-DMA GEMM on operand planes (24 / 25) vs the best of the other kernels, 2 x f16
        P3, P2 = HI | 8, HI | 9
        run("s0 pw1  96->384  M=131072", 32, 64, 96, 384, [(1, 1), (HI | 1, 1), (P3, 1), (P2, 1)], planes=True)
        run("s0 pw2 384->96   M=131072", 32, 64, 384, 96, [(5, 1), (HI | 1, 1), (P3, 1), (P2, 1)], grn=True, planes=True)
        run("s1 pw1 192->768  M=32768 ", 32, 32, 192, 768, [(1, 1), (HI | 2, 1), (P3, 1), (P2, 1)], planes=True)
        run("s1 pw2 768->192  M=32768 ", 32, 32, 768, 192, [(HI | 2, 1), (HI | 2, 2), (P3, 1), (P3, 2), (P2, 2)], grn=True, planes=True)
        run("s2 pw1 384->1536 M=8192  ", 32, 16, 384, 1536, [(HI | 2, 1), (P3, 1), (P2, 1)], planes=True)
        run("s2 pw2 1536->384 M=8192  ", 32, 16, 1536, 384, [(HI | 2, 2), (HI | 10, 1), (HI | 2, 4), (P3, 2), (P3, 4), (P3, 8)], grn=True, planes=True)
        run("s3 pw1 768->3072 M=2048  ", 32, 8, 768, 3072, [(HI | 2, 1), (HI | 1, 1), (HI | 10, 1), (HI | 2, 2), (P3, 1), (P3, 2)], planes=True)
        run("s3 pw2 3072->768 M=2048  ", 32, 8, 3072, 768, [(HI | 2, 8), (HI | 2, 4), (HI | 1, 4), (HI | 10, 2), (HI | 10, 4), (P3, 8), (P3, 16)], grn=True, planes=True)
        run("chunky s2 pw1 1472->5888 M=4096", 16, 16, 1472, 5888, [(HI | 2, 1), (P3, 1)], reps=5, planes=True)
        run("chunky s2 pw2 5888->1472 M=4096", 16, 16, 5888, 1472, [(HI | 2, 2), (P3, 1), (P3, 2)], grn=True, reps=5, planes=True)      # (pc with the GRN A path: K slices of at most 3072)
        sys.exit(0)
    G = [(1, 1), (2, 1), (5, 1), (4, 1), (13, 1), (14, 1)]
    run("s0 pw1  96->384  M=131072", 32, 64, 96, 384, G + [(HI | 1, 1), (HI | 2, 1)])
    run("s0 pw2 384->96   M=131072", 32, 64, 384, 96, G + [(HI | 1, 1)], grn=True)
    run("s1 pw1 192->768  M=32768 ", 32, 32, 192, 768, G + [(HI | 1, 1), (HI | 2, 1)])
    run("s1 pw2 768->192  M=32768 ", 32, 32, 768, 192, G + [(HI | 1, 1), (HI | 2, 1), (HI | 2, 2)], grn=True)
    run("s2 pw1 384->1536 M=8192  ", 32, 16, 384, 1536, G + [(HI | 1, 1), (HI | 2, 1)])
    run("s2 pw2 1536->384 M=8192  ", 32, 16, 1536, 384, G + [(HI | 1, 1), (HI | 2, 1), (HI | 1, 2), (HI | 2, 2), (HI | 2, 4)], grn=True)
    run("s3 pw1 768->3072 M=2048  ", 32, 8, 768, 3072, G + [(HI | 1, 1), (HI | 2, 1), (HI | 2, 2)])
    run("s3 pw2 3072->768 M=2048  ", 32, 8, 3072, 768, G + [(HI | 1, 1), (HI | 2, 1), (HI | 1, 4), (HI | 2, 4), (HI | 2, 8), (HI | 1, 8)], grn=True)
