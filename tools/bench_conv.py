#!/usr/bin/env python
"""Micro-benchmark + accuracy check of vs_conv_gemm variants on the shapes of the VideoSeal path (GPU box only)."""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videoseal_amd import native as N  # noqa: E402
from videoseal_amd.engine import Act, ConvW, HipEngine, pack_conv, rup  # noqa: E402


class Eng(HipEngine):
    def __init__(self):
        self.dev = torch.device("cuda"); self.lib = N.lib(); self._ws = {}; self.kernel_timers = None; self.use_split = True; self.autotune = False; self._tile_cache = {}; self.time_all_convs = False


def bench(eng, name, B, Cin, H, W, Cout, k, variants, reps=10, check=True):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H, W, Cin, generator=g).cuda()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).cuda()
    xa = Act(x, B, H, W, Cin, Cin)
    wt, cp = pack_conv(w, Cin)
    cw = ConvW(wt, None, Cout, k, k, cp).with_blk()
    out = eng.new_act("o", B, H, W, Cout)
    flops = 2.0 * B * H * W * Cout * Cin * k * k
    ref = None
    if check:   # fp64 reference on a slice of frames
        nb = min(B, 2)
        ref = F.conv2d(x[:nb].permute(0, 3, 1, 2).double(), w.double(), padding=k // 2).permute(0, 2, 3, 1).float()
    for vname, hint in variants:
        eng.use_split = not (hint & N.CONV_FORCE_F32)
        for _ in range(2):
            eng.conv(xa, cw, out, pad=k // 2, tile_hint=hint)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            eng.conv(xa, cw, out, pad=k // 2, tile_hint=hint)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        err = ""
        if ref is not None:
            got = out.t.view(B, H, W, out.ld)[:ref.shape[0], ..., :Cout]
            err = f" max|err| vs fp64 = {(got - ref).abs().max().item():.2e} (|ref|max {ref.abs().max().item():.1f})"
        print(f"{name:28s} {vname:14s} {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s{err}", flush=True)


if __name__ == "__main__":
    eng = Eng()
    F32, SP = N.CONV_FORCE_F32, N.CONV_FORCE_SPLIT
    V = [("f32 128x128", F32 | 1), ("split 128x128", SP | 1), ("split 128x64", SP | 2), ("split 128x96", SP | 5),
         ("pc 256x128", 6), ("pc 128x128", 7), ("pc 128x64", 8), ("pc 256x64", 9),
         ("pc 256x128 prod-idle", 6 | 0x400), ("pc 256x128 no-mfma", 6 | 0x800), ("pc 128x128 prod-idle", 7 | 0x400), ("pc 128x128 no-mfma", 7 | 0x800),
         ("pc 256x64 prod-idle", 9 | 0x400), ("pc 256x64 no-mfma", 9 | 0x800)]
    bench(eng, "bottleneck 384->384 32^2 B32", 32, 384, 32, 32, 384, 3, V)
    AB = [("split128x64 full", SP | 2), ("  no-global", SP | 2 | 0x100), ("  no-mfma", SP | 2 | 0x200), ("  no-ldsstore", SP | 2 | 0x400),
          ("  no-global,no-store", SP | 2 | 0x500), ("  only-mfma+ldsread", SP | 2 | 0xD00), ("  no-mfma,no-global", SP | 2 | 0x300),
          ("split128x128 full", SP | 1), ("  no-global", SP | 1 | 0x100), ("  no-mfma", SP | 1 | 0x200), ("  only-mfma+ldsread", SP | 1 | 0xD00),
          ("f32 128x64 full", F32 | 2), ("  no-global", F32 | 2 | 0x100), ("  no-mfma", F32 | 2 | 0x200), ("  only-mfma+ldsread", F32 | 2 | 0xD00)]
    PV = [("split best4w 128x96", SP | 5), ("patch 128x32", 10), ("patch 128x64", 11), ("patch 128x128", 12)]
    bench(eng, "PATCH bottleneck 384 32^2 B32", 32, 384, 32, 32, 384, 3, PV)
    PA = [("patch 128x128", 12), (" no-Bload", 12 | 0x100), (" no-mfma", 12 | 0x200), (" no-Bstore", 12 | 0x400), (" no-Bload,no-Bstore", 12 | 0x500),
          (" only mfma+ldsread", 12 | 0xD00), (" no-mfma,no-Bload", 12 | 0x300), ("patch 128x64", 11), (" no-Bload", 11 | 0x100), (" no-mfma", 11 | 0x200),
          (" only mfma+ldsread", 11 | 0xD00)]
    bench(eng, "PABL bottleneck B32", 32, 384, 32, 32, 384, 3, PA, check=False)
    bench(eng, "PATCH rb16 256^2 B32", 32, 16, 256, 256, 16, 3, [("split 256x32", SP | 3), ("patch 128x32", 10)], check=False)
    bench(eng, "PATCH up2 64->16 256^2 B32", 32, 64, 256, 256, 16, 3, [("split 256x32", SP | 3), ("patch 128x32", 10)], check=False)
    bench(eng, "PATCH rb32 128^2 B32", 32, 32, 128, 128, 32, 3, [("split 256x32", SP | 3), ("patch 128x32", 10)], check=False)
    bench(eng, "PATCH up1 128->32 128^2 B32", 32, 128, 128, 128, 32, 3, [("split 256x32", SP | 3), ("patch 128x32", 10)], check=False)
    bench(eng, "PATCH rb64 64^2 B32", 32, 64, 64, 64, 64, 3, [("split 128x64", SP | 2), ("patch 128x64", 11), ("patch 128x32", 10)], check=False)
    bench(eng, "PATCH up0 768->64 64^2 B32", 32, 768, 64, 64, 64, 3, [("split 128x64", SP | 2), ("patch 128x64", 11), ("patch 128x32", 10)], check=False)
    bench(eng, "PATCH rb128 32^2 B32", 32, 128, 32, 32, 128, 3, [("split 128x128", SP | 1), ("patch 128x128", 12), ("patch 128x64", 11)], check=False)
    bench(eng, "ABLATE bottleneck B32", 32, 384, 32, 32, 384, 3, AB, check=False)
    bench(eng, "bottleneck 384->384 32^2 B8", 8, 384, 32, 32, 384, 3, V)
    bench(eng, "up0 768->64 64^2 B32", 32, 768, 64, 64, 64, 3, [("f32 128x64", F32 | 2), ("split 128x64", SP | 2), ("split 256x32", SP | 3)])
    bench(eng, "pw1 96->384 64^2 B32", 32, 96, 64, 64, 384, 1, V)
    bench(eng, "pw2 384->96 64^2 B32", 32, 384, 64, 64, 96, 1, V)
    bench(eng, "pw1 384->1536 16^2 B32", 32, 384, 16, 16, 1536, 1, V)
    bench(eng, "pw2 3072->768 8^2 B32", 32, 3072, 8, 8, 768, 1, V)
    bench(eng, "rb 16->16 256^2 B32", 32, 16, 256, 256, 16, 3, [("f32 256x32", F32 | 3), ("split 256x32", SP | 3)], check=False)
    bench(eng, "rb 32->32 128^2 B32", 32, 32, 128, 128, 32, 3, [("f32 256x32", F32 | 3), ("split 256x32", SP | 3)], check=False)
    bench(eng, "rb 64->64 64^2 B32", 32, 64, 64, 64, 64, 3, [("f32 128x64", F32 | 2), ("split 128x64", SP | 2)], check=False)
