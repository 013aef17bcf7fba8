#!/usr/bin/env python
"""CPU check (float64, torch autograd as the judge) of the index / adjoint formulas the backward kernels are written to:
reflect-padded 3x3 patch-matrix adjoint, flipped depthwise taps and their weight gradient, stride-2 conv adjoint through zero dilation,
bilinear x2 adjoint in gather form, BatchNorm(batch statistics) + ReLU, GELU + GRN, and the (anti-aliased) bilinear resize of
csrc/shell.hip::make_taps with its gather-form adjoint (the kernel for it is still to be written).  Runs anywhere: python tools/check_bwd_formulas.py"""
import math

import torch
import torch.nn.functional as F

dt = torch.float64
torch.manual_seed(0)


def refl_srcs(y, k, H):
    o, o1 = [], y - k + 1
    if 0 <= o1 < H:
        o.append(o1)
    if k == 0 and y == 1:
        o.append(0)
    if k == 2 and y == H - 2:
        o.append(H - 1)
    return o


def up_contrib(y, H):
    out = []
    for Y in range(max(0, 2 * y - 1), min(2 * H, 2 * y + 3)):
        src = max((Y + 0.5) / 2 - 0.5, 0.0)
        i0 = int(src)
        i1 = min(i0 + 1, H - 1)
        f = src - i0
        w = (1 - f if i0 == y else 0.0) + (f if i1 == y else 0.0)
        if w != 0:
            out.append((Y, w))
    return out


def tri(x):
    x = abs(x)
    return 1 - x if x < 1 else 0.0


def resize_taps(i, n_in, n_out, aa):
    """csrc/shell.hip::make_taps restated: (first source index, normalised weights) of output index i"""
    scale = n_in / n_out
    if aa:
        support = scale if scale >= 1 else 1.0
        center, inv = scale * (i + 0.5), (1 / scale if scale >= 1 else 1.0)
        lo, hi = max(int(center - support + 0.5), 0), min(int(center + support + 0.5), n_in)
        ws = [tri((j + lo - center + 0.5) * inv) for j in range(hi - lo)]
        tot = sum(ws)
        return lo, [w / tot for w in ws]
    src = max(scale * (i + 0.5) - 0.5, 0.0)
    i0 = min(int(src), n_in - 1)
    return (i0, [1 - (src - i0), src - i0]) if i0 < n_in - 1 else (i0, [1.0])


def resize_candidates(s, n_in, n_out, aa):
    """outputs whose taps can include source s (the loop bounds of the gather-form adjoint kernel still to be written)"""
    scale = n_in / n_out
    support = (scale if scale >= 1 else 1.0) if aa else 1.0
    return max(math.floor((s + 0.5 - support - 1) / scale - 0.5), 0), min(math.ceil((s + 0.5 + support + 1) / scale - 0.5), n_out - 1)


def main():
    worst = 0.0
    for aa in (True, False):                                            # (anti-aliased) bilinear resize: forward taps and gather-form adjoint
        for (n_in, n_out) in [(64, 88), (64, 72), (90, 64), (70, 64), (64, 64), (5, 17), (17, 5), (3, 2), (2, 3)]:
            x = torch.randn(1, 1, 1, n_in, dtype=dt, requires_grad=True)
            y = F.interpolate(x, size=(1, n_out), mode="bilinear", align_corners=False, antialias=aa)
            dy = torch.randn_like(y)
            y.backward(dy)
            for o in range(n_out):
                lo, ws = resize_taps(o, n_in, n_out, aa)
                worst = max(worst, abs(float(sum(w * x.detach()[0, 0, 0, lo + j] for j, w in enumerate(ws)) - y.detach()[0, 0, 0, o])))
            for s in range(n_in):
                a, b = resize_candidates(s, n_in, n_out, aa)
                acc = 0.0
                for o in range(a, b + 1):
                    lo, ws = resize_taps(o, n_in, n_out, aa)
                    if lo <= s < lo + len(ws):
                        acc += ws[s - lo] * float(dy[0, 0, 0, o])
                worst = max(worst, abs(acc - float(x.grad[0, 0, 0, s])))
    for (H, W) in [(5, 7), (2, 2), (3, 9), (2, 5)]:                     # col2im, reflect
        C, N = 2, 3
        x = torch.randn(1, C, H, W, dtype=dt, requires_grad=True)
        w = torch.randn(N, C, 3, 3, dtype=dt)
        y = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w)
        dy = torch.randn_like(y)
        y.backward(dy)
        dcols = torch.einsum("nhw,ntc->hwtc", dy[0], w.permute(0, 2, 3, 1).reshape(N, 9, C))
        dx = torch.zeros(H, W, C, dtype=dt)
        for yy in range(H):
            for xx in range(W):
                for ky in range(3):
                    for kx in range(3):
                        for oy in refl_srcs(yy, ky, H):
                            for ox in refl_srcs(xx, kx, W):
                                dx[yy, xx] += dcols[oy, ox, ky * 3 + kx]
        worst = max(worst, float((dx - x.grad[0].permute(1, 2, 0)).abs().max()))
    for (H, W) in [(8, 8), (9, 7), (5, 6)]:                             # stride-2 conv adjoint
        x = torch.randn(2, 3, H, W, dtype=dt, requires_grad=True)
        w = torch.randn(4, 3, 3, 3, dtype=dt)
        y = F.conv2d(x, w, stride=2, padding=1)
        dy = torch.randn_like(y)
        y.backward(dy)
        dil = torch.zeros(2, 4, H, W, dtype=dt)
        dil[:, :, ::2, ::2][:, :, : y.shape[2], : y.shape[3]] = dy
        worst = max(worst, float((F.conv2d(dil, w.flip(2, 3).permute(1, 0, 2, 3), padding=1) - x.grad).abs().max()))
    for (H, W) in [(4, 5), (1, 3), (2, 2), (7, 3)]:                     # bilinear x2 adjoint
        x = torch.randn(1, 2, H, W, dtype=dt, requires_grad=True)
        y = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
        dy = torch.randn_like(y)
        y.backward(dy)
        dx = torch.zeros(H, W, 2, dtype=dt)
        for yy in range(H):
            for xx in range(W):
                for (Y, wy) in up_contrib(yy, H):
                    for (X, wx) in up_contrib(xx, W):
                        dx[yy, xx] += wy * wx * dy[0, :, Y, X]
        worst = max(worst, float((dx - x.grad[0].permute(1, 2, 0)).abs().max()))
    # depthwise 7x7: flipped taps, weight gradient index ranges
    C, H, W = 3, 6, 5
    x = torch.randn(1, C, H, W, dtype=dt, requires_grad=True)
    w = torch.randn(C, 1, 7, 7, dtype=dt, requires_grad=True)
    y = F.conv2d(x, w, padding=3, groups=C)
    dy = torch.randn_like(y)
    y.backward(dy)
    worst = max(worst, float((F.conv2d(dy, w.detach().reshape(C, 49).flip(1).reshape(C, 1, 7, 7), padding=3, groups=C) - x.grad).abs().max()))
    dw = torch.zeros(49, C, dtype=dt)
    for t in range(49):
        ky, kx = t // 7, t % 7
        for yy in range(H):
            iy = yy + ky - 3
            if iy < 0 or iy >= H:
                continue
            for xx in range(3 - kx if kx < 3 else 0, min(W + 3 - kx, W)):
                dw[t] += dy[0, :, yy, xx] * x.detach()[0, :, iy, xx + kx - 3]
    worst = max(worst, float((dw - w.grad.reshape(C, 49).t()).abs().max()))
    # BatchNorm (batch statistics) + ReLU
    rows, C = 50, 6
    raw = torch.randn(rows, C, dtype=dt, requires_grad=True)
    g_, b_ = torch.randn(C, dtype=dt, requires_grad=True), torch.randn(C, dtype=dt, requires_grad=True)
    y = F.relu(F.batch_norm(raw, None, None, g_, b_, training=True, eps=1e-5))
    dy = torch.randn_like(y)
    y.backward(dy)
    with torch.no_grad():
        r = raw.detach()
        mu, var = r.mean(0), r.var(0, unbiased=False)
        rstd = 1 / torch.sqrt(var + 1e-5)
        xh = (r - mu) * rstd
        g = dy * ((g_.detach() * xh + b_.detach()) > 0)
        s0, s1 = (g * xh).sum(0), g.sum(0)
        draw = g_.detach() * rstd * (g - s1 / rows - xh * s0 / rows)
    worst = max(worst, float((draw - raw.grad).abs().max()), float((s0 - g_.grad).abs().max()), float((s1 - b_.grad).abs().max()))
    # GELU + GRN
    B, HW, C = 3, 20, 8
    h1 = torch.randn(B, HW, C, dtype=dt, requires_grad=True)
    gamma, beta = torch.randn(C, dtype=dt, requires_grad=True), torch.randn(C, dtype=dt, requires_grad=True)
    h2 = F.gelu(h1)
    G = torch.norm(h2, p=2, dim=1, keepdim=True)
    h3 = gamma * (h2 * (G / (G.mean(dim=-1, keepdim=True) + 1e-6))) + beta + h2
    d3 = torch.randn_like(h3)
    h3.backward(d3)
    with torch.no_grad():
        h2d, Gd = h2.detach(), G.detach()[:, 0]
        M = Gd.mean(-1, keepdim=True) + 1e-6
        nx = Gd / M
        s, t = (d3 * h2d).sum(1), d3.sum(1)
        dnx = s * gamma.detach()
        dG = dnx / M - (dnx * Gd).sum(-1, keepdim=True) / (C * M ** 2)
        xx = h1.detach()
        gp = 0.5 * (1 + torch.erf(xx / math.sqrt(2))) + xx * torch.exp(-0.5 * xx * xx) / math.sqrt(2 * math.pi)
        dh1 = (d3 * (gamma.detach() * nx + 1)[:, None, :] + h2d * (dG / Gd)[:, None, :]) * gp
    worst = max(worst, float((dh1 - h1.grad).abs().max()), float(((s * nx).sum(0) - gamma.grad).abs().max()), float((t.sum(0) - beta.grad).abs().max()))
    print(f"worst absolute deviation from autograd over all formulas: {worst:.2e}")
    assert worst < 1e-12


if __name__ == "__main__":
    main()
