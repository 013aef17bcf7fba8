"""The detector fine-tuning step on the HIP path (first slice of the training backward).

train.py:517-523 freezes the embedder for its fine-tuning epochs (`lambda_i = lambda_d = 0`, `balanced = False`): the step of
train.py:626-643 is then  forward -> decoding loss (videosealloss.py:150-156) -> backward through the extractor only.  `DetectorStep`
is that step for the ConvNeXt-V2 extractor (extractor.py:154-167, convnext.py:41-57, pixel_decoder.py:61-83): a forward that keeps what
the backward needs, the loss, and the gradient of every `detector.*` parameter accumulated into `.grad` (so the reference's torch
optimizers and schedulers apply unchanged).

Kernels: backward-DATA products reuse the forward GEMM kernels on transposed weights with the exact 3 x bf16 operand split (full fp32
exponent range for the gradients); everything else is csrc/bwd_ops.hip.  No CPU path: without the library or a GPU this raises.
Pinned against the reference's own `loss.backward()` (tests/golden/make_golden_bwd.py -> tests/test_gpu_bwd.py)."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import native as N
from .engine import A_MUL_GRN, Act, ConvW, HipEngine, pack_conv, rup

BWD_ARITH = 3          # 3 x bf16: exact operand split, fp32 exponent range (vs_conv_desc_t::arith)


class DetectorStep:
    """`step(imgs_aug, msgs)`: one accumulation step of train.py:626-643 with the embedder frozen.

    imgs_aug: [B, 3, S, S] in [0, 1] on the model's device (what `model(imgs, masks, msgs)["imgs_aug"]` returns: already at img_size);
    msgs: [B, k] or [1, k] bits.  Returns (loss, logits); gradients are ADDED to `.grad` of the detector's parameters, scaled by
    `grad_scale` (= 1 / accumulation_steps in train.py:641)."""

    def __init__(self, model):
        if model.embedder.cfg.extractor == "sam":
            raise N.NativeError("DetectorStep covers the ConvNeXt-V2 extractor of the released 1.0 / PixelSeal / ChunkySeal cards")
        if model.embedder.cfg.stem_stride != 4:
            raise N.NativeError("DetectorStep needs the non-overlapping 4x4 stride-4 stem (VideoSeal 1.0 / PixelSeal)")
        self.model = model
        self._ones: Dict[int, torch.Tensor] = {}

    # ------------------------------------------------------------------ small helpers
    def _act(self, eng: HipEngine, tag: str, B, H, W, Cc, ld=None) -> Act:
        ld = ld or rup(Cc, 4)
        return Act(eng.buf("tr." + tag, B * H * W * ld, zero=True), B, H, W, Cc, ld)

    def _vec(self, eng, n: int, value: float) -> torch.Tensor:
        key = (n, value)
        if key not in self._ones:
            self._ones[key] = torch.full((n,), value, device=eng.dev, dtype=torch.float32)
        return self._ones[key]

    def _gelu(self, eng, x: Act, out: Act):
        n = rup(x.C, 4)
        N.check(eng.lib.vs_scale_shift_act(N.ptr(x.t), x.rows, n, x.ld, N.ptr(self._vec(eng, x.ld, 1.0)), N.ptr(self._vec(eng, x.ld, 0.0)),
                                           N.ACT_GELU, None, 0, N.ptr(out.t), out.ld, N.stream()), "vs_scale_shift_act")
        return out

    def _colsum(self, eng, x: Act, n: int) -> torch.Tensor:
        """sum over the rows of the first n columns (bias gradients): the fp64 column sums of the BatchNorm kernels"""
        L = eng.lib
        part = eng.buf("tr.cs.part", 2 * int(L.vs_bn_partial_doubles(x.rows, x.ld)))
        sums = eng.buf("tr.cs.sums", 2 * (2 * x.ld + 2)).view(torch.float64)[: 2 * x.ld + 1]
        N.check(L.vs_bn_partial_sums(N.ptr(x.t), x.rows, x.C, x.ld, N.ptr(part), N.ptr(sums), N.stream()), "vs_bn_partial_sums")
        return sums[:n].float()

    def _wgrad(self, eng, dy: Act, n: int, x: Act, k: int) -> torch.Tensor:
        """dW[n][k] = sum_rows dy[row][:n]^T x[row][:k]"""
        L = eng.lib
        part = eng.buf("tr.wg.part", int(L.vs_gemm_wgrad_partial_floats(dy.rows, n, k)))
        dw = torch.empty(n, k, device=eng.dev, dtype=torch.float32)
        N.check(L.vs_gemm_wgrad(N.ptr(dy.t), dy.ld, n, N.ptr(x.t), x.ld, k, dy.rows, N.ptr(part), N.ptr(dw), N.stream()), "vs_gemm_wgrad")
        return dw

    def _ln_bwd(self, eng, x: Act, dy: Act, w: torch.Tensor, tag: str) -> Tuple[Act, torch.Tensor, torch.Tensor]:
        L = eng.lib
        dx = self._act(eng, tag, x.B, x.H, x.W, x.C, x.ld)
        stats = eng.buf("tr.ln.stats", 2 * x.rows)
        part = eng.buf("tr.ln.part", int(L.vs_colreduce_partial_floats(1, x.rows, x.ld)))
        dw = torch.empty(x.C, device=eng.dev, dtype=torch.float32)
        db = torch.empty(x.C, device=eng.dev, dtype=torch.float32)
        N.check(L.vs_layernorm_bwd(N.ptr(x.t), x.ld, N.ptr(dy.t), dy.ld, N.ptr(w), x.rows, x.C, 1e-6, N.ptr(dx.t), dx.ld, N.ptr(stats), N.ptr(part),
                                   N.ptr(dw), N.ptr(db), N.stream()), "vs_layernorm_bwd")
        return dx, dw, db

    @staticmethod
    def _tw(weight2d: torch.Tensor, in_ld: int) -> ConvW:
        """the transposed matrix of a Linear / 1x1 layer as a forward GEMM weight: dX = dY W"""
        wt = weight2d.float().t().contiguous()                    # [K_out = fan_in][fan_out]
        p, cp = pack_conv(wt[:, :, None, None], in_ld)
        return ConvW(p, None, wt.shape[0], 1, 1, cp)

    # ------------------------------------------------------------------ forward that keeps the backward's operands
    def _forward(self, eng: HipEngine, x: Act):
        if eng.X is None:
            eng._pack_extractor(eng._g)
        c, X, L, st = eng.cfg, eng.X, eng.lib, N.stream()
        d, B = c.dims, x.B
        xld = eng._xld
        S = {"x": x, "stages": []}
        Ho, Wo = x.H // 4, x.W // 4
        t = self._act(eng, "stem.c", B, Ho, Wo, d[0])
        eng.conv(x, X["stem"], t, geom=(Wo, 16, 16, 4, 1, 0, 0))
        cur = self._act(eng, "st0.in", B, Ho, Wo, d[0], xld(d[0]))
        eng.layernorm(t, X["stem_ln"][0], X["stem_ln"][1], cur)
        S["stem_pre"] = t
        for sti in range(4):
            rec = {"blocks": []}
            if sti > 0:
                dn = X["down"][sti - 1]
                ln = self._act(eng, f"st{sti}.dln", B, cur.H, cur.W, cur.C, cur.ld)
                eng.layernorm(cur, dn["lnw"], dn["lnb"], ln)
                Ho, Wo = cur.H // 2, cur.W // 2
                nxt = self._act(eng, f"st{sti}.in", B, Ho, Wo, d[sti], xld(d[sti]))
                eng.conv(ln, dn["conv"], nxt, geom=(Wo, 2 * ln.ld, 2 * ln.ld, 2, 1, 0, 0))
                rec["down_in"], rec["down_ln"] = cur, ln
                cur = nxt
            Cc, HW = d[sti], cur.H * cur.W
            ld4 = xld(4 * Cc)
            for j, blk in enumerate(X["stages"][sti]):
                tg = f"st{sti}.b{j}."
                t0 = self._act(eng, tg + "t0", B, cur.H, cur.W, Cc, cur.ld)
                N.check(L.vs_dwconv7(N.ptr(cur.t), B, cur.H, cur.W, Cc, cur.ld, N.ptr(blk["wdw"]), N.ptr(blk["bdw"]), 0, None, 0, N.ptr(t0.t), t0.ld,
                                     st), "vs_dwconv7")
                u = self._act(eng, tg + "u", B, cur.H, cur.W, Cc, cur.ld)
                eng.layernorm(t0, blk["lnw"], blk["lnb"], u)
                h1 = self._act(eng, tg + "h1", B, cur.H, cur.W, 4 * Cc, ld4)
                eng.conv(u, blk["pw1"], h1)
                h3 = self._act(eng, tg + "h3", B, cur.H, cur.W, 4 * Cc, ld4)
                self._gelu(eng, h1, h3)
                part = eng.buf("tr.grn.part", ((HW + 63) // 64) * B * 4 * Cc)
                scale = eng.buf("tr.grn.scale", B * ld4 + 16)
                N.check(L.vs_grn_scale(N.ptr(h3.t), B, HW, 4 * Cc, ld4, N.ptr(blk["gamma"]), N.ptr(part), N.ptr(scale), st), "vs_grn_scale")
                N.check(L.vs_grn_apply(N.ptr(h3.t), B, HW, 4 * Cc, ld4, N.ptr(scale), ld4, N.ptr(blk["beta"]), st), "vs_grn_apply")
                out = self._act(eng, tg + "out", B, cur.H, cur.W, Cc, cur.ld)
                eng.conv(h3, blk["pw2"], out, res=cur, a_mul=A_MUL_GRN)
                rec["blocks"].append(dict(x=cur, t0=t0, u=u, h1=h1, h3=h3))
                cur = out
            S["stages"].append(rec)
        # pixel decoder (upscale_stages [1]): reflect-pad conv3x3 as patch matrix + GEMM, LayerNorm, GELU, mean, Linear
        Cl = d[-1]
        g = eng._g
        wh = g("detector.pixel_decoder.output_upscaling.0.upsample_block.2.weight").float()            # [Cl, Cl, 3, 3]
        wcols = torch.zeros(Cl, 9, cur.ld, device=eng.dev)
        wcols[:, :, :Cl] = wh.permute(0, 2, 3, 1).reshape(Cl, 9, Cl)
        wcols = wcols.reshape(Cl, 9 * cur.ld)
        S["head_wcols"] = wcols
        pw, cpw = pack_conv(wcols[:, :, None, None], 9 * cur.ld)
        cols = Act(eng.buf("tr.head.cols", cur.rows * 9 * cur.ld, zero=True), B, cur.H, cur.W, 9 * cur.ld, 9 * cur.ld)
        N.check(L.vs_im2col3x3(N.ptr(cur.t), B, cur.H, cur.W, cur.ld, N.PAD_REFLECT, N.ptr(cols.t), st), "vs_im2col3x3")
        hc = self._act(eng, "head.c", B, cur.H, cur.W, Cl)
        eng.conv(cols, ConvW(pw, None, Cl, 1, 1, cpw), hc)
        z = self._act(eng, "head.z", B, cur.H, cur.W, Cl)
        eng.layernorm(hc, X["head_ln"][0], X["head_ln"][1], z)
        hl = self._act(eng, "head.l", B, cur.H, cur.W, Cl)
        self._gelu(eng, z, hl)
        logits = torch.empty(B, c.nbits + 1, device=eng.dev, dtype=torch.float32)
        N.check(L.vs_pool_linear(N.ptr(hl.t), B, hl.H * hl.W, hl.C, hl.ld, N.ptr(X["lin_w"]), N.ptr(X["lin_b"]), c.nbits + 1, N.ptr(logits), st),
                "vs_pool_linear")
        S.update(last=cur, cols=cols, hc=hc, z=z, hl=hl)
        return logits, S

    # ------------------------------------------------------------------ backward
    def _backward(self, eng: HipEngine, S, dlogits: torch.Tensor) -> Dict[str, torch.Tensor]:
        c, X, L, st, g = eng.cfg, eng.X, eng.lib, N.stream(), eng._g
        d = c.dims
        G: Dict[str, torch.Tensor] = {}
        pd, cn = "detector.pixel_decoder", "detector.convnext"
        hl, z, hc, cols, cur = S["hl"], S["z"], S["hc"], S["cols"], S["last"]
        B, HW, Cl, N1 = hl.B, hl.H * hl.W, hl.C, c.nbits + 1
        # ---- Linear on the pooled features
        pooled = eng.buf("tr.head.pooled", B * hl.ld)
        N.check(L.vs_colmean(N.ptr(hl.t), B, HW, hl.ld, N.ptr(pooled), st), "vs_colmean")
        dlt = dlogits.t().contiguous()                                    # [N1][B]
        dlw = torch.empty(N1, Cl, device=eng.dev, dtype=torch.float32)
        N.check(L.vs_matmul_small(N.ptr(dlt), B, N.ptr(pooled), hl.ld, N1, Cl, B, N.ptr(dlw), Cl, st), "vs_matmul_small")
        dlb = torch.empty(N1, device=eng.dev, dtype=torch.float32)
        N.check(L.vs_matmul_small(N.ptr(self._vec(eng, B, 1.0)), B, N.ptr(dlogits), N1, 1, N1, B, N.ptr(dlb), N1, st), "vs_matmul_small")
        G[pd + ".linear.weight"], G[pd + ".linear.bias"] = dlw, dlb
        dpooled = eng.buf("tr.head.dpooled", B * hl.ld)
        N.check(L.vs_matmul_small(N.ptr(dlogits), N1, N.ptr(X["lin_w"]), Cl, B, Cl, N1, N.ptr(dpooled), hl.ld, st), "vs_matmul_small")
        # ---- mean over (H, W), GELU, LayerNorm
        dz = self._act(eng, "head.dz", B, hl.H, hl.W, Cl)
        N.check(L.vs_pool_gelu_bwd(N.ptr(z.t), z.ld, N.ptr(dpooled), hl.ld, B, HW, Cl, N.ptr(dz.t), dz.ld, st), "vs_pool_gelu_bwd")
        dhc, dw, db = self._ln_bwd(eng, hc, dz, X["head_ln"][0], "head.dhc")
        G[pd + ".output_upscaling.0.upsample_block.3.weight"], G[pd + ".output_upscaling.0.upsample_block.3.bias"] = dw, db
        # ---- reflect-pad conv3x3 (no bias)
        dwc = self._wgrad(eng, dhc, Cl, cols, 9 * cur.ld)
        G[pd + ".output_upscaling.0.upsample_block.2.weight"] = dwc.view(Cl, 3, 3, cur.ld)[..., :Cl].permute(0, 3, 1, 2).contiguous()
        dcols = Act(eng.buf("tr.head.dcols", cur.rows * 9 * cur.ld, zero=True), B, cur.H, cur.W, 9 * cur.ld, 9 * cur.ld)
        eng.conv(dhc, self._tw(S["head_wcols"], dhc.ld), dcols, arith=BWD_ARITH)
        dy = self._act(eng, "st3.dy", B, cur.H, cur.W, cur.C, cur.ld)
        N.check(L.vs_col2im3x3_reflect(N.ptr(dcols.t), B, cur.H, cur.W, cur.ld, N.ptr(dy.t), st), "vs_col2im3x3_reflect")
        # ---- stages, last to first
        for sti in (3, 2, 1, 0):
            rec = S["stages"][sti]
            Cc = d[sti]
            for j in range(len(rec["blocks"]) - 1, -1, -1):
                sv, p = rec["blocks"][j], f"{cn}.stages.{sti}.{j}"
                blk = X["stages"][sti][j]
                xin, t0, u, h1, h3 = sv["x"], sv["t0"], sv["u"], sv["h1"], sv["h3"]
                Bc, H, W, HWc, ld4 = xin.B, xin.H, xin.W, xin.H * xin.W, h1.ld
                tg = f"st{sti}."
                # pwconv2
                G[p + ".pwconv2.weight"] = self._wgrad(eng, dy, Cc, h3, 4 * Cc)
                G[p + ".pwconv2.bias"] = self._colsum(eng, dy, Cc)
                d3 = self._act(eng, tg + "d3", Bc, H, W, 4 * Cc, ld4)
                eng.conv(dy, self._tw(g(p + ".pwconv2.weight"), dy.ld), d3, arith=BWD_ARITH)
                # GRN + GELU
                dh1 = self._act(eng, tg + "dh1", Bc, H, W, 4 * Cc, ld4)
                part = eng.buf("tr.grn.bpart", int(L.vs_colreduce_partial_floats(Bc, HWc, ld4)))
                coef = eng.buf("tr.grn.coef", 6 * Bc * ld4)
                dgam = torch.empty(4 * Cc, device=eng.dev, dtype=torch.float32)
                dbet = torch.empty(4 * Cc, device=eng.dev, dtype=torch.float32)
                N.check(L.vs_gelu_grn_bwd(N.ptr(h1.t), ld4, N.ptr(d3.t), ld4, N.ptr(blk["gamma"]), Bc, HWc, 4 * Cc, N.ptr(part), N.ptr(coef),
                                          N.ptr(dh1.t), ld4, N.ptr(dgam), N.ptr(dbet), st), "vs_gelu_grn_bwd")
                G[p + ".grn.gamma"], G[p + ".grn.beta"] = dgam.view(1, 1, 1, -1), dbet.view(1, 1, 1, -1)
                # pwconv1
                G[p + ".pwconv1.weight"] = self._wgrad(eng, dh1, 4 * Cc, u, Cc)
                G[p + ".pwconv1.bias"] = self._colsum(eng, dh1, 4 * Cc)
                du = self._act(eng, tg + "du", Bc, H, W, Cc, xin.ld)
                eng.conv(dh1, self._tw(g(p + ".pwconv1.weight"), ld4), du, arith=BWD_ARITH)
                # LayerNorm
                dt0, dw, db = self._ln_bwd(eng, t0, du, blk["lnw"], tg + "dt0")
                G[p + ".norm.weight"], G[p + ".norm.bias"] = dw, db
                # depthwise 7x7 (+ the residual branch)
                dwp = eng.buf("tr.dw.part", int(L.vs_dwconv7_wgrad_partial_floats(Bc, H, xin.ld)))
                dwd = torch.empty(49, xin.ld, device=eng.dev, dtype=torch.float32)
                N.check(L.vs_dwconv7_wgrad(N.ptr(xin.t), xin.ld, N.ptr(dt0.t), dt0.ld, Bc, H, W, Cc, N.ptr(dwp), N.ptr(dwd), st), "vs_dwconv7_wgrad")
                G[p + ".dwconv.weight"] = dwd[:, :Cc].t().reshape(Cc, 1, 7, 7).contiguous()
                G[p + ".dwconv.bias"] = self._colsum(eng, dt0, Cc)
                dx = self._act(eng, tg + f"dx{j & 1}", Bc, H, W, Cc, xin.ld)
                N.check(L.vs_dwconv7(N.ptr(dt0.t), Bc, H, W, Cc, dt0.ld, N.ptr(blk["wdw"]), None, 1, N.ptr(dy.t), dy.ld, N.ptr(dx.t), dx.ld, st),
                        "vs_dwconv7")
                dy = dx
            if sti > 0:        # downsample layer: LayerNorm(cf) -> conv 2x2 stride 2
                dn, p = X["down"][sti - 1], f"{cn}.downsample_layers.{sti}"
                cin_act, ln = rec["down_in"], rec["down_ln"]
                Cin, Bc = d[sti - 1], ln.B
                CP = dn["conv"].CinP                                                # rup(2 * ln.ld, 16)
                patches = Act(eng.buf("tr.dn.patches", dy.rows * 2 * CP, zero=True), Bc, dy.H, dy.W, 2 * CP, 2 * CP)
                N.check(L.vs_patchify(N.ptr(ln.t), Bc, ln.H, ln.W, ln.ld, 2, N.ptr(patches.t), st), "vs_patchify")
                dwp = self._wgrad(eng, dy, Cc, patches, 2 * CP)                     # [Cout][ky * CP + kx * ld + c]
                G[p + ".1.weight"] = dwp.view(Cc, 2, CP)[:, :, : 2 * ln.ld].reshape(Cc, 2, 2, ln.ld)[..., :Cin].permute(0, 3, 1, 2).contiguous()
                G[p + ".1.bias"] = self._colsum(eng, dy, Cc)
                dcols = Act(eng.buf("tr.dn.dcols", dy.rows * 2 * CP, zero=True), Bc, dy.H, dy.W, 2 * CP, 2 * CP)
                eng.conv(dy, self._tw(dn["conv"].wt, dy.ld), dcols, arith=BWD_ARITH)
                dln = self._act(eng, f"st{sti}.g_dln", Bc, ln.H, ln.W, Cin, ln.ld)
                N.check(L.vs_unpatch(N.ptr(dcols.t), Bc, ln.H, ln.W, ln.ld, 2, N.ptr(dln.t), st), "vs_unpatch")
                dy, dw, db = self._ln_bwd(eng, cin_act, dln, dn["lnw"], f"st{sti}.dcur")
                G[p + ".0.weight"], G[p + ".0.bias"] = dw, db
        # ---- stem: conv 4x4 stride 4 -> LayerNorm(cf); the frames themselves receive no gradient (embedder frozen)
        p = f"{cn}.downsample_layers.0"
        t, x = S["stem_pre"], S["x"]
        dt, dw, db = self._ln_bwd(eng, t, dy, X["stem_ln"][0], "stem.dt")
        G[p + ".1.weight"], G[p + ".1.bias"] = dw, db
        patches = Act(eng.buf("tr.stem.patches", dt.rows * 64, zero=True), x.B, dt.H, dt.W, 64, 64)
        N.check(L.vs_patchify(N.ptr(x.t), x.B, x.H, x.W, 4, 4, N.ptr(patches.t), st), "vs_patchify")
        dws = self._wgrad(eng, dt, d[0], patches, 64)
        G[p + ".0.weight"] = dws.view(d[0], 4, 4, 4)[..., :3].permute(0, 3, 1, 2).contiguous()
        G[p + ".0.bias"] = self._colsum(eng, dt, d[0])
        return G

    # ------------------------------------------------------------------ public
    def step(self, imgs_aug: torch.Tensor, msgs: torch.Tensor, temperature: float = 1.0, grad_scale: float = 1.0,
             accumulate: bool = True):
        model = self.model
        eng = model._engine()
        with torch.cuda.device(eng.dev):
            x = N.f32c(imgs_aug.to(eng.dev))
            if tuple(x.shape[-2:]) != (model.img_size, model.img_size):
                raise ValueError(f"imgs_aug must be at the extractor's working size {model.img_size} (forward() returns it resized)")
            rgb, _ = eng.resize_pre(x, (x.shape[-2], x.shape[-1]), False, want_rgb=True, mul=2.0, add=-1.0, tag="tr.det.in")
            logits, S = self._forward(eng, rgb)
            B, k = logits.shape[0], logits.shape[1] - 1
            m = msgs.to(eng.dev).to(torch.int32).contiguous()
            if m.dim() != 2 or m.shape[1] != k or m.shape[0] not in (1, B):
                raise ValueError(f"msgs must be [{B} or 1, {k}]")
            dlogits = torch.empty_like(logits)
            loss = torch.empty(1, device=eng.dev, dtype=torch.float32)
            N.check(eng.lib.vs_bce_logits(N.ptr(logits), N.ptr(m), m.shape[0], B, k, float(temperature), float(grad_scale), N.ptr(dlogits),
                                          N.ptr(loss), N.stream()), "vs_bce_logits")
            grads = self._backward(eng, S, dlogits)
        if accumulate:
            params = dict(model.named_parameters())
            for name, gten in grads.items():
                prm = params[name]
                gten = gten.reshape(prm.shape)
                if prm.grad is None:
                    prm.grad = gten.clone()
                else:
                    prm.grad.add_(gten)
        return loss[0], logits, grads
