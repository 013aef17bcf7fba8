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

import os
from typing import Dict, List, Optional, Tuple

import torch

from . import native as N
from .engine import A_MUL_GRN, Act, ConvW, HipEngine, pack_conv, pack_conv_bwd, rup

DIRECT_WGRAD = os.environ.get("VIDEOSEAL_DIRECT_WGRAD", "1") != "0"       # 0: 3x3 weight gradients through the explicit patch matrix (A/B)
BWD_ARITH = 3          # 3 x bf16: exact operand split, fp32 exponent range (vs_conv_desc_t::arith)


class _GradSink(dict):
    """gradient dictionary that drops everything when only the data path is wanted (the adaptive-weight probes of videosealloss.py:72-107
    differentiate the loss with respect to one layer: no parameter gradient of the extractor is needed there)"""

    def __init__(self, keep: bool = True):
        super().__init__()
        self.keep = keep

    def __setitem__(self, k, v):
        if self.keep:
            super().__setitem__(k, v)


class DetectorStep:
    """`step(imgs_aug, msgs)`: one accumulation step of train.py:626-643 with the embedder frozen.

    imgs_aug: [B, 3, S, S] in [0, 1] on the model's device (what `model(imgs, masks, msgs)["imgs_aug"]` returns: already at img_size);
    msgs: [B, k] or [1, k] bits.  Returns (loss, logits); gradients are ADDED to `.grad` of the detector's parameters, scaled by
    `grad_scale` (= 1 / accumulation_steps in train.py:641)."""

    def __init__(self, model):
        self.model = model
        self.vit = model.embedder.cfg.extractor == "sam"       # the legacy card's SAM-style ViT (vit.py:14-144): _forward_vit / _backward_vit
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
        if getattr(self, "_skip_w", False):
            return None
        L = eng.lib
        part = eng.buf("tr.cs.part", 2 * int(L.vs_bn_partial_doubles(x.rows, x.ld)))
        sums = eng.buf("tr.cs.sums", 2 * (2 * x.ld + 2)).view(torch.float64)[: 2 * x.ld + 1]
        N.check(L.vs_bn_partial_sums(N.ptr(x.t), x.rows, x.C, x.ld, N.ptr(part), N.ptr(sums), N.stream()), "vs_bn_partial_sums")
        return sums[:n].float()

    def _wgrad(self, eng, dy: Act, n: int, x: Act, k: int) -> torch.Tensor:
        """dW[n][k] = sum_rows dy[row][:n]^T x[row][:k]"""
        if getattr(self, "_skip_w", False):
            return None
        L = eng.lib
        part = eng.buf("tr.wg.part", int(L.vs_gemm_wgrad_partial_floats(dy.rows, n, k)))
        dw = torch.empty(n, k, device=eng.dev, dtype=torch.float32)
        N.check(L.vs_gemm_wgrad(N.ptr(dy.t), dy.ld, n, N.ptr(x.t), x.ld, k, dy.rows, N.ptr(part), N.ptr(dw), N.stream()), "vs_gemm_wgrad")
        return dw

    def _ln_bwd(self, eng, x: Act, dy: Act, w: torch.Tensor, tag: str, eps: float = 1e-6) -> Tuple[Act, torch.Tensor, torch.Tensor]:
        L = eng.lib
        dx = self._act(eng, tag, x.B, x.H, x.W, x.C, x.ld)
        stats = eng.buf("tr.ln.stats", 2 * x.rows)
        part = eng.buf("tr.ln.part", int(L.vs_colreduce_partial_floats(1, x.rows, x.ld)))
        dw = torch.empty(x.C, device=eng.dev, dtype=torch.float32)
        db = torch.empty(x.C, device=eng.dev, dtype=torch.float32)
        N.check(L.vs_layernorm_bwd(N.ptr(x.t), x.ld, N.ptr(dy.t), dy.ld, N.ptr(w), x.rows, x.C, eps, N.ptr(dx.t), dx.ld, N.ptr(stats), N.ptr(part),
                                   N.ptr(dw), N.ptr(db), N.stream()), "vs_layernorm_bwd")
        return dx, dw, db

    @staticmethod
    def _tw(weight2d: torch.Tensor, in_ld: int) -> ConvW:
        """the transposed matrix of a Linear / 1x1 layer as a forward GEMM weight: dX = dY W"""
        p, cp = pack_conv_bwd(weight2d[:, :, None, None], in_ld)   # [K_out = fan_in][fan_out]: one launch (csrc/pack.hip)
        return ConvW(p, None, weight2d.shape[1], 1, 1, cp)

    # ------------------------------------------------------------------ forward that keeps the backward's operands
    def _forward(self, eng: HipEngine, x: Act):
        """the training forward runs on the exact, range-free 3 x bf16 split: weights drift during training, and an activation that leaves the
        f16 range of the fast arithmetic would only show up as a NaN loss"""
        eng.arith = BWD_ARITH
        if self.vit:
            return self._forward_vit(eng, x)
        if eng.X is None:
            eng._pack_extractor(eng._g)
        c, X, L, st = eng.cfg, eng.X, eng.lib, N.stream()
        d, B = c.dims, x.B
        xld = eng._xld
        S = {"x": x, "stages": []}
        ss = c.stem_stride                      # 4 (VideoSeal 1.0 / PixelSeal) or 2 (ChunkySeal: overlapping 4 x 4 patches, convnext.py:109)
        Ho, Wo = (x.H - 4) // ss + 1, (x.W - 4) // ss + 1
        t = self._act(eng, "stem.c", B, Ho, Wo, d[0])
        eng.conv(x, X["stem"], t, geom=(Wo, ss * 4, 16, ss, 1, 0, 0))
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
        logits = self._head_forward(eng, cur, X, S)
        return logits, S

    # ------------------------------------------------------------------ the legacy card's SAM-style ViT extractor (vit.py:14-144)
    def _forward_vit(self, eng: HipEngine, x: Act):
        """engine.vit_extractor_forward with every operand of the backward kept: per block the input of each LayerNorm, qkv, the attention
        output, the MLP's pre-activation and its GELU; the neck's four maps."""
        if eng.X is None:
            eng._pack_vit(eng._g)
        c, V, L, st, B = eng.cfg, eng.X, eng.lib, N.stream(), x.B
        P, D_ = c.vit_patch, c.vit_dim
        gh, gw = x.H // P, x.W // P
        if gh * gw * D_ != V["pos"].numel():
            raise N.NativeError(f"ViT extractor: {x.H}x{x.W} input does not match the position table")
        pos = V["pos"].repeat(B, 1).contiguous()
        tok = self._act(eng, "vit.x", B, gh, gw, D_)
        eng.conv(x, V["patch"], tok, geom=(gw, P * 4, P * 4, P, 1, 0, 0), res=Act(pos, B, gh, gw, D_, D_))
        hd, hid = D_ // c.vit_heads, int(D_ * c.vit_mlp_ratio)
        S = {"x": x, "blocks": [], "grid": (gh, gw)}
        for i, blk in enumerate(V["blocks"]):
            tg = f"vit.b{i}."
            n1 = self._act(eng, tg + "n1", B, gh, gw, D_)
            eng.layernorm(tok, blk["n1"][0], blk["n1"][1], n1, eps=1e-5)               # nn.LayerNorm default eps
            qkv = self._act(eng, tg + "qkv", B, gh, gw, 3 * D_)
            eng.conv(n1, blk["qkv"], qkv)
            att = self._act(eng, tg + "att", B, gh, gw, D_)
            N.check(L.vs_vit_attention(N.ptr(qkv.t), B, gh, gw, c.vit_heads, hd, blk["window"], N.ptr(blk["rel_h"]), N.ptr(blk["rel_w"]),
                                       N.ptr(att.t), st), "vs_vit_attention")
            mid = self._act(eng, tg + "tokm", B, gh, gw, D_)
            eng.conv(att, blk["proj"], mid, res=tok)                                   # x = shortcut + proj(attn)
            n2 = self._act(eng, tg + "n2", B, gh, gw, D_)
            eng.layernorm(mid, blk["n2"][0], blk["n2"][1], n2, eps=1e-5)
            z1 = self._act(eng, tg + "z1", B, gh, gw, hid)
            eng.conv(n2, blk["lin1"], z1)
            a1 = self._act(eng, tg + "a1", B, gh, gw, hid)
            self._gelu(eng, z1, a1)
            out = self._act(eng, tg + "out", B, gh, gw, D_)
            eng.conv(a1, blk["lin2"], out, res=mid)                                    # x = x + mlp(norm2(x))
            S["blocks"].append(dict(x=tok, n1=n1, qkv=qkv, att=att, mid=mid, n2=n2, z1=z1, a1=a1))
            tok = out
        n0 = self._act(eng, "vit.neck0", B, gh, gw, c.vit_out)
        eng.conv(tok, V["neck0"], n0)
        n1 = self._act(eng, "vit.neck1", B, gh, gw, c.vit_out, eng._xld(c.vit_out))
        eng.layernorm(n0, V["neck1"][0], V["neck1"][1], n1)
        n2 = self._act(eng, "vit.neck2", B, gh, gw, c.vit_out)
        eng.conv(n1, V["neck2"], n2, pad=1)
        n3 = self._act(eng, "vit.neck3", B, gh, gw, c.vit_out, eng._xld(c.vit_out))
        eng.layernorm(n2, V["neck3"][0], V["neck3"][1], n3)
        S.update(tok_last=tok, n0=n0, n1=n1, n2=n2)
        logits = self._head_forward(eng, n3, V, S)
        return logits, S

    def _backward_vit(self, eng: HipEngine, S, dlogits: torch.Tensor, want_params: bool = True, want_input: bool = False):
        """gradients of every `detector.*` parameter of the ViT extractor (vit.py:55-127 neck / blocks / patch embedding, 302-360 attention with
        its relative-position tables) and / or of the input frames"""
        c, V, L, st, g = eng.cfg, eng.X, eng.lib, N.stream(), eng._g
        G: Dict[str, torch.Tensor] = _GradSink(want_params)
        self._skip_w = not want_params
        ie = "detector.image_encoder"
        D_, O_ = c.vit_dim, c.vit_out
        hd, hid = D_ // c.vit_heads, int(D_ * c.vit_mlp_ratio)
        gh, gw = S["grid"]
        B = S["x"].B
        dn3 = self._head_backward(eng, S, dlogits, V, G, want_params)
        # ---- neck: conv1x1 -> LayerNorm2d -> conv3x3 (zero padding 1) -> LayerNorm2d, no biases (vit.py:104-121)
        n0, n1, n2, tok = S["n0"], S["n1"], S["n2"], S["tok_last"]
        dn2, dw, db = self._ln_bwd(eng, n2, dn3, V["neck3"][0], "vit.g.dn2")
        G[ie + ".neck.3.weight"], G[ie + ".neck.3.bias"] = dw, db
        if want_params:
            part = eng.buf("tr.wg.part", int(L.vs_conv3x3_wgrad_partial_floats(O_, n1.ld, B, gh, gw, 1)))
            dwn = torch.empty(O_, 9 * n1.ld, device=eng.dev, dtype=torch.float32)
            if L.vs_conv3x3_wgrad_supported(O_, n1.ld, 1):
                N.check(L.vs_conv3x3_wgrad(N.ptr(dn2.t), dn2.ld, O_, N.ptr(n1.t), n1.ld, B, gh, gw, 1, N.PAD_ZERO, N.ptr(part), N.ptr(dwn), st),
                        "vs_conv3x3_wgrad")
            else:
                cols = Act(eng.buf("tr.vit.cols", n1.rows * 9 * n1.ld, zero=True), B, gh, gw, 9 * n1.ld, 9 * n1.ld)
                N.check(L.vs_im2col3x3(N.ptr(n1.t), B, gh, gw, n1.ld, N.PAD_ZERO, N.ptr(cols.t), st), "vs_im2col3x3")
                dwn = self._wgrad(eng, dn2, O_, cols, 9 * n1.ld)
            G[ie + ".neck.2.weight"] = dwn.view(O_, 3, 3, n1.ld)[..., :O_].permute(0, 3, 1, 2).contiguous()
        dn1 = self._act(eng, "vit.g.dn1", B, gh, gw, O_, n1.ld)
        eng.conv(dn2, EmbedderBackward._flip_t(g(ie + ".neck.2.weight"), dn2.ld), dn1, pad=1, arith=BWD_ARITH)
        dn0, dw, db = self._ln_bwd(eng, n0, dn1, V["neck1"][0], "vit.g.dn0")
        G[ie + ".neck.1.weight"], G[ie + ".neck.1.bias"] = dw, db
        w0 = g(ie + ".neck.0.weight").reshape(O_, D_)
        G[ie + ".neck.0.weight"] = self._wgrad(eng, dn0, O_, tok, D_)
        if want_params:
            G[ie + ".neck.0.weight"] = G[ie + ".neck.0.weight"].view(O_, D_, 1, 1)
        dtok = self._act(eng, "vit.g.dtok", B, gh, gw, D_)
        eng.conv(dn0, self._tw(w0, dn0.ld), dtok, arith=BWD_ARITH)
        # ---- blocks, last to first (vit.py:146-193)
        for i in range(len(S["blocks"]) - 1, -1, -1):
            sv, blk, p = S["blocks"][i], V["blocks"][i], f"{ie}.blocks.{i}"
            tg = f"vit.g.b{i & 1}."
            # x = mid + lin2(gelu(lin1(norm2(mid))))
            G[p + ".mlp.lin2.weight"] = self._wgrad(eng, dtok, D_, sv["a1"], hid)
            G[p + ".mlp.lin2.bias"] = self._colsum(eng, dtok, D_)
            da1 = self._act(eng, tg + "da1", B, gh, gw, hid)
            eng.conv(dtok, self._tw(g(p + ".mlp.lin2.weight"), dtok.ld), da1, arith=BWD_ARITH)
            dz1 = self._act(eng, tg + "dz1", B, gh, gw, hid)
            N.check(L.vs_gelu_bwd(N.ptr(sv["z1"].t), sv["z1"].ld, N.ptr(da1.t), da1.ld, da1.rows, hid, N.ptr(dz1.t), dz1.ld, st), "vs_gelu_bwd")
            G[p + ".mlp.lin1.weight"] = self._wgrad(eng, dz1, hid, sv["n2"], D_)
            G[p + ".mlp.lin1.bias"] = self._colsum(eng, dz1, hid)
            dn2b = self._act(eng, tg + "dn2", B, gh, gw, D_)
            eng.conv(dz1, self._tw(g(p + ".mlp.lin1.weight"), dz1.ld), dn2b, arith=BWD_ARITH)
            dmid_ln, dw, db = self._ln_bwd(eng, sv["mid"], dn2b, blk["n2"][0], tg + "dmidln", eps=1e-5)
            G[p + ".norm2.weight"], G[p + ".norm2.bias"] = dw, db
            dmid = self._act(eng, tg + "dmid", B, gh, gw, D_)
            torch.add(dtok.t, dmid_ln.t, out=dmid.t)                                   # the residual branch + the MLP branch
            # mid = x + proj(attention(qkv(norm1(x))))
            G[p + ".attn.proj.weight"] = self._wgrad(eng, dmid, D_, sv["att"], D_)
            G[p + ".attn.proj.bias"] = self._colsum(eng, dmid, D_)
            datt = self._act(eng, tg + "datt", B, gh, gw, D_)
            eng.conv(dmid, self._tw(g(p + ".attn.proj.weight"), dmid.ld), datt, arith=BWD_ARITH)
            dqkv = self._act(eng, tg + "dqkv", B, gh, gw, 3 * D_)
            scr = eng.buf("tr.vit.attn.scratch", int(L.vs_vit_attention_bwd_scratch_floats(B, gh, gw, c.vit_heads, blk["window"])))
            rel = blk["rel_h"] is not None
            drh = torch.empty_like(blk["rel_h"]) if (rel and want_params) else None
            drw = torch.empty_like(blk["rel_w"]) if (rel and want_params) else None
            N.check(L.vs_vit_attention_bwd(N.ptr(sv["qkv"].t), N.ptr(sv["att"].t), N.ptr(datt.t), B, gh, gw, c.vit_heads, hd, blk["window"],
                                           N.ptr(blk["rel_h"]), N.ptr(blk["rel_w"]), N.ptr(dqkv.t), N.ptr(scr), N.ptr(drh), N.ptr(drw), st),
                    "vs_vit_attention_bwd")
            if rel:
                G[p + ".attn.rel_pos_h"], G[p + ".attn.rel_pos_w"] = drh, drw
            G[p + ".attn.qkv.weight"] = self._wgrad(eng, dqkv, 3 * D_, sv["n1"], D_)
            G[p + ".attn.qkv.bias"] = self._colsum(eng, dqkv, 3 * D_)
            dn1b = self._act(eng, tg + "dn1", B, gh, gw, D_)
            eng.conv(dqkv, self._tw(g(p + ".attn.qkv.weight"), dqkv.ld), dn1b, arith=BWD_ARITH)
            dx_ln, dw, db = self._ln_bwd(eng, sv["x"], dn1b, blk["n1"][0], tg + "dxln", eps=1e-5)
            G[p + ".norm1.weight"], G[p + ".norm1.bias"] = dw, db
            dtok = self._act(eng, tg + "dx", B, gh, gw, D_)
            torch.add(dmid.t, dx_ln.t, out=dtok.t)
        # ---- patch embedding (conv P x P stride P + bias) + absolute positions (vit.py:66-69, 129-131)
        x = S["x"]
        P = c.vit_patch
        CP = rup(P * x.ld, 16)
        if want_params:
            G[ie + ".pos_embed"] = dtok.t.view(B, gh * gw * D_).sum(0).view(1, gh, gw, D_) if B > 1 else dtok.t.view(1, gh, gw, D_).clone()
            patches = Act(eng.buf("tr.vit.patches", dtok.rows * P * CP, zero=True), B, gh, gw, P * CP, P * CP)
            N.check(L.vs_patchify_s(N.ptr(x.t), B, x.H, x.W, x.ld, P, P, N.ptr(patches.t), st), "vs_patchify_s")
            dwp = self._wgrad(eng, dtok, D_, patches, P * CP)                               # [D][ky * CP + kx * ld + c]
            G[ie + ".patch_embed.proj.weight"] = dwp.view(D_, P, CP)[:, :, : P * x.ld].reshape(D_, P, P, x.ld)[..., :3].permute(0, 3, 1, 2).contiguous()
        G[ie + ".patch_embed.proj.bias"] = self._colsum(eng, dtok, D_)
        if not want_input:
            return G
        dcols = Act(eng.buf("tr.vit.dcols", dtok.rows * P * CP, zero=True), B, gh, gw, P * CP, P * CP)
        eng.conv(dtok, self._tw(V["patch"].wt, dtok.ld), dcols, arith=BWD_ARITH)
        drgb = eng.buf("tr.stem.drgb", x.rows * 4, zero=True)
        N.check(L.vs_unpatch_s(N.ptr(dcols.t), B, x.H, x.W, x.ld, P, P, N.ptr(drgb), st), "vs_unpatch_s")
        dimg = torch.empty(B, 3, x.H, x.W, device=eng.dev, dtype=torch.float32)
        N.check(L.vs_nhwc_to_nchw_scaled(N.ptr(drgb), B, x.H, x.W, 3, 4, 2.0, N.ptr(dimg), st), "vs_nhwc_to_nchw_scaled")
        return G, dimg

    def _head_forward(self, eng: HipEngine, cur: Act, X, S) -> torch.Tensor:
        """pixel decoder (pixel_decoder.py:61-83, upscale_stages [1]): reflect-pad conv3x3 as patch matrix + GEMM, LayerNorm, GELU, mean, Linear"""
        c, L, st, B = eng.cfg, eng.lib, N.stream(), cur.B
        Cl = cur.C
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
        return logits

    def _head_backward(self, eng: HipEngine, S, dlogits: torch.Tensor, X, G, want_params: bool) -> Act:
        """pixel decoder backward: fills the `detector.pixel_decoder.*` gradients, returns the gradient of its input map"""
        c, L, st = eng.cfg, eng.lib, N.stream()
        pd = "detector.pixel_decoder"
        hl, z, hc, cols, cur = S["hl"], S["z"], S["hc"], S["cols"], S["last"]
        B, HW, Cl, N1 = hl.B, hl.H * hl.W, hl.C, c.nbits + 1
        # ---- Linear on the pooled features
        pooled = eng.buf("tr.head.pooled", B * hl.ld)
        N.check(L.vs_colmean(N.ptr(hl.t), B, HW, hl.ld, N.ptr(pooled), st), "vs_colmean")
        if want_params:
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
        if want_params:
            dwc = self._wgrad(eng, dhc, Cl, cols, 9 * cur.ld)
            G[pd + ".output_upscaling.0.upsample_block.2.weight"] = dwc.view(Cl, 3, 3, cur.ld)[..., :Cl].permute(0, 3, 1, 2).contiguous()
        dcols = Act(eng.buf("tr.head.dcols", cur.rows * 9 * cur.ld, zero=True), B, cur.H, cur.W, 9 * cur.ld, 9 * cur.ld)
        eng.conv(dhc, self._tw(S["head_wcols"], dhc.ld), dcols, arith=BWD_ARITH)
        dy = self._act(eng, "st3.dy", B, cur.H, cur.W, cur.C, cur.ld)
        N.check(L.vs_col2im3x3_reflect(N.ptr(dcols.t), B, cur.H, cur.W, cur.ld, N.ptr(dy.t), st), "vs_col2im3x3_reflect")
        return dy

    # ------------------------------------------------------------------ backward
    def _backward(self, eng: HipEngine, S, dlogits: torch.Tensor, want_params: bool = True, want_input: bool = False):
        """gradients of every `detector.*` parameter (want_params) and / or of the extractor's input frames [B, 3, S, S] in [0, 1]
        (want_input: the path the generator-side loss takes back to the embedder).  Returns G, or (G, d_input) with want_input."""
        if self.vit:
            return self._backward_vit(eng, S, dlogits, want_params, want_input)
        c, X, L, st, g = eng.cfg, eng.X, eng.lib, N.stream(), eng._g
        d = c.dims
        G: Dict[str, torch.Tensor] = _GradSink(want_params)
        self._skip_w = not want_params          # _wgrad / _colsum return None: only the data path runs
        cn = "detector.convnext"
        dy = self._head_backward(eng, S, dlogits, X, G, want_params)
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
                if want_params:
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
                if want_params:
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
        if want_params:
            patches = Act(eng.buf("tr.stem.patches", dt.rows * 64, zero=True), x.B, dt.H, dt.W, 64, 64)
            N.check(L.vs_patchify_s(N.ptr(x.t), x.B, x.H, x.W, 4, 4, c.stem_stride, N.ptr(patches.t), st), "vs_patchify_s")
            dws = self._wgrad(eng, dt, d[0], patches, 64)
            G[p + ".0.weight"] = dws.view(d[0], 4, 4, 4)[..., :3].permute(0, 3, 1, 2).contiguous()
        G[p + ".0.bias"] = self._colsum(eng, dt, d[0])
        if not want_input:
            return G
        # backward-data of the stem: d patches = dt W  ([rows][64], k = ky * 16 + kx * 4 + c), un-patched to NHWC, then back to frame planes
        # times d(x * 2 - 1) / dx (extractor.py:163)
        dcols = Act(eng.buf("tr.stem.dcols", dt.rows * 64, zero=True), x.B, dt.H, dt.W, 64, 64)
        eng.conv(dt, self._tw(X["stem"].wt, dt.ld), dcols, arith=BWD_ARITH)
        drgb = eng.buf("tr.stem.drgb", x.rows * 4, zero=True)
        N.check(L.vs_unpatch_s(N.ptr(dcols.t), x.B, x.H, x.W, 4, 4, c.stem_stride, N.ptr(drgb), st), "vs_unpatch_s")
        dimg = torch.empty(x.B, 3, x.H, x.W, device=eng.dev, dtype=torch.float32)
        N.check(L.vs_nhwc_to_nchw_scaled(N.ptr(drgb), x.B, x.H, x.W, 3, 4, 2.0, N.ptr(dimg), st), "vs_nhwc_to_nchw_scaled")
        return G, dimg

    # ------------------------------------------------------------------ public
    def step(self, imgs_aug: torch.Tensor, msgs: torch.Tensor, temperature: float = 1.0, grad_scale: float = 1.0,
             accumulate: bool = True):
        model = self.model
        eng = model._engine()
        eng.begin_training_pass()
        # this pass rewrites the 'tr.*' operands an earlier model(...) graph of the detector would read in its backward: make that backward
        # raise (autograd._check_generation) instead of silently using overwritten operands
        model._train_gen["detector"] = model._train_gen.get("detector", 0) + 1
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


class EmbedderBackward:
    """Backward of the U-Net embedder under model.train() (unet.py:17-197 with BatchNorm on batch statistics): gradients of every
    `embedder.*` parameter from d(delta), the gradient with respect to the embedder's output [B, out_ch, S, S] -- the generator side of
    train.py:626-643 up to the JND / blend / augmentation adjoints that produce d(delta) from the loss.

    Called by videoseal_amd.autograd.EmbedTrainFn (the differentiable `Wam.forward`); tests/test_gpu_bwd_unet.py holds the unit tests of its
    kernels and the comparison of every gradient with autograd through the oracle."""

    def __init__(self, model):
        cfg = model.embedder.cfg
        self.rms = cfg.unet_norm == "rms"        # the legacy card: ChanRMSNorm instead of BatchNorm (common.py:172-194), no batch statistics
        self.act = N.ACT_SILU if cfg.unet_act == "silu" else N.ACT_RELU
        if not self.rms and self.act != N.ACT_RELU:
            raise N.NativeError("EmbedderBackward: BatchNorm U-Nets are covered with ReLU only")
        self.model = model
        self.h = DetectorStep.__new__(DetectorStep)          # the small helpers (_act, _vec, _colsum, _wgrad, _ln_bwd, _tw)
        self.h._ones = {}
        self.update_running = False
        self.batch_stats = True

    # ------------------------------------------------------------------ pieces
    def _bn_stats(self, eng: HipEngine, raw: Act, bn: dict):
        """batch statistics of `raw` (global over the ranks with SyncBatchNorm): scale, shift, mean, rstd.  With self.update_running (the
        real training forward) running_mean / running_var / num_batches_tracked move exactly like nn.BatchNorm2d's (momentum 0.1, unbiased)."""
        L, st = eng.lib, N.stream()
        if not self.batch_stats:        # embedder.eval() with trainable parameters: BatchNorm is the affine map of its running statistics
            v = torch.zeros(4, raw.ld, device=eng.dev, dtype=torch.float32)
            rstd = torch.rsqrt(bn["rv"].float() + 1e-5)
            v[0, : raw.C] = bn["w"] * rstd
            v[1, : raw.C] = bn["b"] - bn["rm"].float() * v[0, : raw.C]
            v[2, : raw.C], v[3, : raw.C] = bn["rm"].float(), rstd
            return dict(scale=v[0], shift=v[1], mean=v[2], rstd=v[3], frozen=True)
        part = eng.buf("tr.bn.part", 2 * int(L.vs_bn_partial_doubles(raw.rows, raw.ld)))
        sums = eng.buf("tr.bn.sums", 2 * (2 * raw.ld + 2)).view(torch.float64)[: 2 * raw.ld + 1]
        N.check(L.vs_bn_partial_sums(N.ptr(raw.t), raw.rows, raw.C, raw.ld, N.ptr(part), N.ptr(sums), st), "vs_bn_partial_sums")
        if eng.bn_sync is not None:
            eng.bn_sync(sums)
        v = torch.zeros(4, raw.ld, device=eng.dev, dtype=torch.float32)
        upd = self.update_running
        N.check(L.vs_bn_finish_sums(N.ptr(sums), raw.C, raw.ld, N.ptr(bn["w"]), N.ptr(bn["b"]), 1e-5, 0.1, N.ptr(bn["rm"]) if upd else None,
                                    N.ptr(bn["rv"]) if upd else None, N.ptr(v[0]), N.ptr(v[1]), st), "vs_bn_finish_sums")
        if upd:
            bn["nbt"].add_(1)
        N.check(L.vs_bn_mean_rstd(N.ptr(sums), raw.C, raw.ld, 1e-5, N.ptr(v[2]), N.ptr(v[3]), st), "vs_bn_mean_rstd")
        return dict(scale=v[0], shift=v[1], mean=v[2], rstd=v[3])

    def _affine_act(self, eng, x: Act, scale, shift, act, out: Act, add: Optional[Act] = None):
        N.check(eng.lib.vs_scale_shift_act(N.ptr(x.t), x.rows, rup(x.C, 4), x.ld, N.ptr(scale), N.ptr(shift), act, N.ptr(add.t) if add is not None else None,
                                           add.ld if add is not None else 0, N.ptr(out.t), out.ld, N.stream()), "vs_scale_shift_act")
        return out

    def _bn_bwd(self, eng, raw: Act, dy: Act, stt: dict, tag: str):
        """BatchNorm (batch statistics) + ReLU backward: d raw, d gamma, d beta"""
        L, st = eng.lib, N.stream()
        C = raw.C
        ldp = rup(C, 4)
        part = eng.buf("tr.bnb.part", int(L.vs_bn_bwd_partial_floats(raw.rows, ldp)))
        sums = eng.buf("tr.bnb.sums", 2 * (2 * ldp + 2)).view(torch.float64)[: 2 * ldp + 1]
        dg = torch.empty(C, device=eng.dev, dtype=torch.float32)
        db = torch.empty(C, device=eng.dev, dtype=torch.float32)
        N.check(L.vs_bn_relu_bwd_sums(N.ptr(raw.t), raw.ld, N.ptr(dy.t), dy.ld, N.ptr(stt["mean"]), N.ptr(stt["rstd"]), N.ptr(stt["scale"]),
                                      N.ptr(stt["shift"]), 1, raw.rows, C, N.ptr(part), N.ptr(sums), N.ptr(dg), N.ptr(db), st), "vs_bn_relu_bwd_sums")
        if stt.get("frozen"):                      # running statistics: no coupling between the rows, d raw = scale * g
            sums = torch.zeros_like(sums)
            sums[-1] = raw.rows
        elif eng.bn_sync is not None:
            eng.bn_sync(sums)                      # SyncBatchNorm's backward exchange: [sum g xhat | sum g | rows]
        dx = self.h._act(eng, tag, raw.B, raw.H, raw.W, C, raw.ld)
        N.check(L.vs_bn_relu_bwd_apply(N.ptr(raw.t), raw.ld, N.ptr(dy.t), dy.ld, N.ptr(stt["mean"]), N.ptr(stt["rstd"]), N.ptr(stt["scale"]),
                                       N.ptr(stt["shift"]), 1, N.ptr(sums), raw.rows, C, N.ptr(dx.t), dx.ld, st), "vs_bn_relu_bwd_apply")
        return dx, dg, db

    @staticmethod
    def _flip_t(w4: torch.Tensor, in_ld: int) -> ConvW:
        """backward-data weights of a 3x3 stride-1 pad-1 conv: W'[ci][co][ky][kx] = W[co][ci][2 - ky][2 - kx]"""
        p, cp = pack_conv_bwd(w4, in_ld)
        return ConvW(p, None, w4.shape[1], 3, 3, cp)

    def _cols_zero(self, eng, x: Act, stride: int, tag: str) -> Act:
        L, st = eng.lib, N.stream()
        Ho, Wo = (x.H - 1) // stride + 1, (x.W - 1) // stride + 1
        cols = Act(eng.buf("tr." + tag, x.B * Ho * Wo * 9 * x.ld, zero=True), x.B, Ho, Wo, 9 * x.ld, 9 * x.ld)
        if stride == 1:
            N.check(L.vs_im2col3x3(N.ptr(x.t), x.B, x.H, x.W, x.ld, N.PAD_ZERO, N.ptr(cols.t), st), "vs_im2col3x3")
        else:
            N.check(L.vs_im2col3x3_strided(N.ptr(x.t), x.B, x.H, x.W, x.ld, stride, N.ptr(cols.t), st), "vs_im2col3x3_strided")
        return cols

    def _conv3_wgrad(self, eng, dy: Act, co: int, x: Act, ci: int, stride: int = 1, pad_mode: int = N.PAD_ZERO) -> torch.Tensor:
        L = eng.lib
        if DIRECT_WGRAD and L.vs_conv3x3_wgrad_supported(co, x.ld, stride):      # straight from the image (no rows x 9 ld floats of patches)
            part = eng.buf("tr.wg.part", int(L.vs_conv3x3_wgrad_partial_floats(co, x.ld, x.B, x.H, x.W, stride)))
            dw = torch.empty(co, 9 * x.ld, device=eng.dev, dtype=torch.float32)
            N.check(L.vs_conv3x3_wgrad(N.ptr(dy.t), dy.ld, co, N.ptr(x.t), x.ld, x.B, x.H, x.W, stride, pad_mode, N.ptr(part), N.ptr(dw), N.stream()),
                    "vs_conv3x3_wgrad")
        else:
            if pad_mode == N.PAD_REFLECT:
                cols = Act(eng.buf("tr.cols3", x.rows * 9 * x.ld, zero=True), x.B, x.H, x.W, 9 * x.ld, 9 * x.ld)
                N.check(L.vs_im2col3x3(N.ptr(x.t), x.B, x.H, x.W, x.ld, N.PAD_REFLECT, N.ptr(cols.t), N.stream()), "vs_im2col3x3")
            else:
                cols = self._cols_zero(eng, x, stride, "cols3")
            dw = self.h._wgrad(eng, dy, co, cols, 9 * x.ld)                   # [co][tap * ld + c]
        return dw.view(co, 3, 3, x.ld)[..., :ci].permute(0, 3, 1, 2).contiguous()

    # ------------------------------------------------------------------ forward that keeps the backward's operands
    def _rms_act(self, eng, raw: Act, gamma: torch.Tensor, out: Act, add: Optional[Act] = None):
        N.check(eng.lib.vs_rmsnorm_act(N.ptr(raw.t), raw.rows, raw.C, raw.ld, N.ptr(gamma), self.act, N.ptr(add.t) if add is not None else None,
                                       add.ld if add is not None else 0, N.ptr(out.t), out.ld, N.stream()), "vs_rmsnorm_act")
        return out

    def _rms_bwd(self, eng, raw: Act, gamma: torch.Tensor, dy: Act, tag: str):
        """ChanRMSNorm + activation backward: d raw, d gamma [C, 1, 1]"""
        h = self.h
        dx = h._act(eng, tag, raw.B, raw.H, raw.W, raw.C, raw.ld)
        term = h._act(eng, "e.g.rmsterm", raw.B, raw.H, raw.W, raw.C, raw.ld)
        N.check(eng.lib.vs_rmsnorm_act_bwd(N.ptr(raw.t), raw.rows, raw.C, raw.ld, N.ptr(gamma), self.act, N.ptr(dy.t), dy.ld, N.ptr(dx.t), dx.ld,
                                           N.ptr(term.t), term.ld, N.stream()), "vs_rmsnorm_act_bwd")
        dg = h._colsum(eng, term, raw.C)
        return dx, (dg.view(-1, 1, 1) if dg is not None else None)

    def _resblock_keep(self, eng, x: Act, p, tag: str, out: Optional[Act] = None):
        h = self.h
        cout = p["cout"]
        if self.rms:              # silu(rms(conv(silu(rms(conv(x)))))) + res_conv(x): unet.py:17-39 with common.py:118-119, 172-179
            raw0 = h._act(eng, tag + ".raw0", x.B, x.H, x.W, cout)
            eng.conv(x, p["c0"], raw0, pad=1)
            t = self._rms_act(eng, raw0, p["rms"][0], h._act(eng, tag + ".t", x.B, x.H, x.W, cout))
            raw1 = h._act(eng, tag + ".raw1", x.B, x.H, x.W, cout)
            eng.conv(t, p["c1"], raw1, pad=1)
            rs = h._act(eng, tag + ".rs", x.B, x.H, x.W, cout)
            eng.conv(x, p["res"], rs)
            if out is None:
                out = h._act(eng, tag + ".o", x.B, x.H, x.W, cout)
            self._rms_act(eng, raw1, p["rms"][1], out, add=rs)
            return out, dict(x=x, raw0=raw0, t=t, raw1=raw1)
        raw0 = h._act(eng, tag + ".raw0", x.B, x.H, x.W, cout)
        eng.conv(x, p["c0"], raw0, pad=1)
        s0 = self._bn_stats(eng, raw0, p["bn"][0])
        t = h._act(eng, tag + ".t", x.B, x.H, x.W, cout)
        self._affine_act(eng, raw0, s0["scale"], s0["shift"], N.ACT_RELU, t)
        raw1 = h._act(eng, tag + ".raw1", x.B, x.H, x.W, cout)
        eng.conv(t, p["c1"], raw1, pad=1)
        s1 = self._bn_stats(eng, raw1, p["bn"][1])
        rs = h._act(eng, tag + ".rs", x.B, x.H, x.W, cout)
        eng.conv(x, p["res"], rs)
        if out is None:
            out = h._act(eng, tag + ".o", x.B, x.H, x.W, cout)
        self._affine_act(eng, raw1, s1["scale"], s1["shift"], N.ACT_RELU, out, add=rs)
        return out, dict(x=x, raw0=raw0, t=t, raw1=raw1, s0=s0, s1=s1)

    def forward_keep(self, eng: HipEngine, x: Act, msgs_i32: torch.Tensor, update_running: bool = False):
        """the train-mode forward of the U-Net that keeps every operand of the backward; update_running: BatchNorm's running statistics
        follow (a training step), off for gradient checks that must leave the module untouched.  BatchNorm follows `embedder.training`."""
        self.batch_stats = bool(self.model.embedder.training)
        self.update_running = update_running and self.batch_stats
        eng.arith = BWD_ARITH                  # exact, range-free arithmetic for the training forward (see DetectorStep._forward)
        if eng.Et is None:
            eng._pack_embedder(eng._g, train=True)
        c, E, L, st, g, h = eng.cfg, eng.Et, eng.lib, N.stream(), eng._g, self.h
        B, nlev = x.B, len(c.zc) - 1
        S = {"downs": [], "bott": [], "ups": []}
        o, S["inc"] = self._resblock_keep(eng, x, E["inc"], "e.inc")
        hid = [o]
        for i in range(nlev):
            src = hid[-1]
            Ho, Wo = (src.H - 1) // 2 + 1, (src.W - 1) // 2 + 1
            dwn = h._act(eng, f"e.down{i}.d", B, Ho, Wo, c.zc[i + 1])
            eng.conv(src, E["downs"][i]["down"], dwn, stride=2, pad=1)
            out = h._act(eng, "e.h3", B, Ho, Wo, c.bott) if i == nlev - 1 else None
            if out is not None:
                out = Act(out.t, B, Ho, Wo, c.zc[-1], out.ld)               # the resblock writes columns [0, zc[-1]) of [lat | msg]
            o, rec = self._resblock_keep(eng, dwn, E["downs"][i]["rb"], f"e.down{i}", out=out)
            S["downs"].append(dict(src=src, dwn=dwn, rb=rec))
            hid.append(Act(o.t, B, Ho, Wo, c.bott, o.ld) if i == nlev - 1 else o)
        h3 = hid[-1]
        Bm = msgs_i32.shape[0]
        lat = eng.buf("tr.e.lat", Bm * c.hidden)
        N.check(L.vs_msg_latent(N.ptr(E["table"]), N.ptr(msgs_i32), Bm, c.nbits, c.hidden, N.ptr(lat), st), "vs_msg_latent")
        N.check(L.vs_broadcast_channels(N.ptr(lat), Bm, c.hidden, N.ptr(h3.t), B, h3.H * h3.W, h3.ld, c.zc[-1], st), "vs_broadcast_channels")
        xcur = h3
        for j in range(c.num_blocks):
            xcur, rec = self._resblock_keep(eng, xcur, E["bott"][j], f"e.bott{j}")
            S["bott"].append(rec)
        zz = c.zc[:-1] + [c.bott]
        for k in range(nlev):
            skip = hid.pop()
            up = E["ups"][k]
            cout = zz[nlev - 1 - k]
            cat = h._act(eng, f"e.up{k}.cat", B, 2 * xcur.H, 2 * xcur.W, xcur.C + skip.C)
            N.check(L.vs_upcat2x(N.ptr(xcur.t), xcur.C, xcur.ld, N.ptr(skip.t), skip.C, skip.ld, 2 ** -0.5, B, xcur.H, xcur.W, N.ptr(cat.t), cat.ld, st),
                    "vs_upcat2x")
            wup = g(f"embedder.unet.ups.{k}.up.upsample_block.2.weight").float()              # [cout, cin, 3, 3]; reflection padding (unet.py:181)
            pw, cpw = pack_conv(wup, cat.ld)
            cv = h._act(eng, f"e.up{k}.cv", B, cat.H, cat.W, cout)
            eng.conv(cat, ConvW(pw, None, cout, 3, 3, cpw), cv, pad=1, pad_mode=N.PAD_REFLECT)
            z = h._act(eng, f"e.up{k}.z", B, cat.H, cat.W, cout)
            eng.layernorm(cv, up["lnw"], up["lnb"], z)
            ln = h._act(eng, f"e.up{k}.ln", B, cat.H, cat.W, cout)
            self._affine_act(eng, z, h._vec(eng, z.ld, 1.0), h._vec(eng, z.ld, 0.0), self.act, ln)        # unet.py:61-62: the U-Net's act_layer
            xin = xcur
            xcur, rec = self._resblock_keep(eng, ln, up["rb"], f"e.up{k}")
            S["ups"].append(dict(xin=xin, skip=skip, cat=cat, cv=cv, z=z, rb=rec, cout=cout))
        delta = torch.empty(B * c.out_ch * xcur.H * xcur.W, device=eng.dev, dtype=torch.float32)
        N.check(L.vs_outc_tanh(N.ptr(xcur.t), xcur.H * xcur.W, B, xcur.C, xcur.ld, N.ptr(E["outc_w"]), N.ptr(E["outc_b"]), c.out_ch,
                               1 if c.last_tanh else 0, N.ptr(delta), st), "vs_outc_tanh")
        S.update(last=xcur, delta=delta, msgs=msgs_i32, h3=h3)
        return delta.view(B, c.out_ch, xcur.H, xcur.W), S

    # ------------------------------------------------------------------ backward
    def _resblock_bwd(self, eng, rec, p, name: str, dout: Act, G: Dict[str, torch.Tensor], tag: str, need_dx: bool = True,
                      dx_add: Optional[Act] = None) -> Optional[Act]:
        """gradients of one ResnetBlock (unet.py:17-39); returns d x (+ dx_add) or None"""
        h, g = self.h, eng._g
        x, raw0, t, raw1 = rec["x"], rec["raw0"], rec["t"], rec["raw1"]
        cin, cout = x.C, raw1.C
        if self.rms:
            draw1, dg = self._rms_bwd(eng, raw1, p["rms"][1], dout, tag + ".draw1")
            G[name + ".double_conv.4.gamma"] = dg
        else:
            draw1, dg, db = self._bn_bwd(eng, raw1, dout, rec["s1"], tag + ".draw1")
            G[name + ".double_conv.4.weight"], G[name + ".double_conv.4.bias"] = dg, db
        G[name + ".res_conv.weight"] = h._wgrad(eng, dout, cout, x, cin).view(cout, cin, 1, 1)
        G[name + ".res_conv.bias"] = h._colsum(eng, dout, cout)
        G[name + ".double_conv.3.weight"] = self._conv3_wgrad(eng, draw1, cout, t, cout)
        dt = h._act(eng, tag + ".dt", x.B, x.H, x.W, cout)
        eng.conv(draw1, self._flip_t(g(name + ".double_conv.3.weight"), draw1.ld), dt, pad=1, arith=BWD_ARITH)
        if self.rms:
            draw0, dg = self._rms_bwd(eng, raw0, p["rms"][0], dt, tag + ".draw0")
            G[name + ".double_conv.1.gamma"] = dg
        else:
            draw0, dg, db = self._bn_bwd(eng, raw0, dt, rec["s0"], tag + ".draw0")
            G[name + ".double_conv.1.weight"], G[name + ".double_conv.1.bias"] = dg, db
        G[name + ".double_conv.0.weight"] = self._conv3_wgrad(eng, draw0, cout, x, cin)
        if not need_dx:
            return None
        dxr = h._act(eng, tag + ".dxr", x.B, x.H, x.W, cin, x.ld)             # through the 1x1 res_conv (+ whatever else flows into x)
        eng.conv(dout, h._tw(g(name + ".res_conv.weight").reshape(cout, cin), dout.ld), dxr, arith=BWD_ARITH, res=dx_add)
        dx = h._act(eng, tag + ".dx", x.B, x.H, x.W, cin, x.ld)
        eng.conv(draw0, self._flip_t(g(name + ".double_conv.0.weight"), draw0.ld), dx, pad=1, arith=BWD_ARITH, res=dxr)
        return dx

    def backward(self, eng: HipEngine, S, ddelta: torch.Tensor, outc_only: bool = False) -> Dict[str, torch.Tensor]:
        """outc_only: stop after the output convolution -- all `get_last_layer()` needs (the adaptive-weight probes of videosealloss.py:86-90)"""
        c, E, L, st, g, h = eng.cfg, eng.Et, eng.lib, N.stream(), eng._g, self.h
        u = "embedder.unet"
        G: Dict[str, torch.Tensor] = {}
        nlev = len(c.zc) - 1
        last = S["last"]
        B, HW = last.B, last.H * last.W
        # ---- output conv (+ tanh)
        dd = N.f32c(ddelta.to(eng.dev)).reshape(-1)
        dx = h._act(eng, "e.g.outc", B, last.H, last.W, last.C, last.ld)
        dv = Act(eng.buf("tr.e.g.dv", last.rows * 4, zero=True), B, last.H, last.W, c.out_ch, 4)
        N.check(L.vs_outc_tanh_bwd(N.ptr(S["delta"]), N.ptr(dd), HW, B, last.C, N.ptr(E["outc_w"]), c.out_ch, 1 if c.last_tanh else 0, N.ptr(dx.t),
                                   dx.ld, N.ptr(dv.t), st), "vs_outc_tanh_bwd")
        G[u + ".outc.weight"] = h._wgrad(eng, dv, c.out_ch, last, last.C).view(c.out_ch, last.C, 1, 1)
        G[u + ".outc.bias"] = h._colsum(eng, dv, c.out_ch)
        if outc_only:
            return G
        dcur = dx
        dskips: Dict[int, Act] = {}                     # gradient that reaches hid[i] through its skip connection
        # ---- up path, last group first
        for k in range(nlev - 1, -1, -1):
            rec, up = S["ups"][k], E["ups"][k]
            name = f"{u}.ups.{k}"
            dln = self._resblock_bwd(eng, rec["rb"], up["rb"], name + ".conv", dcur, G, f"e.g.up{k}")
            z, cv, cat, cout = rec["z"], rec["cv"], rec["cat"], rec["cout"]
            dz = h._act(eng, f"e.g.up{k}.dz", B, z.H, z.W, cout)
            N.check(L.vs_act_bwd(N.ptr(z.t), z.ld, N.ptr(dln.t), dln.ld, z.rows, cout, self.act, N.ptr(dz.t), dz.ld, st), "vs_act_bwd")
            dcv, dw, db = h._ln_bwd(eng, cv, dz, up["lnw"], f"e.g.up{k}.dcv")
            G[name + ".up.upsample_block.3.weight"], G[name + ".up.upsample_block.3.bias"] = dw, db
            wname = name + ".up.upsample_block.2.weight"
            G[wname] = self._conv3_wgrad(eng, dcv, cout, cat, cat.C, pad_mode=N.PAD_REFLECT)
            # backward data of the reflection-padded conv: the zero-padded transposed conv over the PADDED map, folded back onto the image
            canvas = h._act(eng, f"e.g.up{k}.canvas", B, cat.H + 2, cat.W + 2, cout, dcv.ld)
            N.check(L.vs_pad_embed1(N.ptr(dcv.t), B, cat.H, cat.W, dcv.ld, N.ptr(canvas.t), st), "vs_pad_embed1")
            dxp = h._act(eng, f"e.g.up{k}.dxp", B, cat.H + 2, cat.W + 2, cat.C, cat.ld)
            eng.conv(canvas, self._flip_t(g(wname), canvas.ld), dxp, pad=1, arith=BWD_ARITH)
            dcat = h._act(eng, f"e.g.up{k}.dcat", B, cat.H, cat.W, cat.C, cat.ld)
            N.check(L.vs_reflect_fold1(N.ptr(dxp.t), B, cat.H, cat.W, cat.ld, N.ptr(dcat.t), st), "vs_reflect_fold1")
            xin, skip = rec["xin"], rec["skip"]
            dxin = h._act(eng, f"e.g.up{k}.dxin", B, xin.H, xin.W, xin.C, xin.ld)
            dsk = h._act(eng, f"e.g.up{k}.dskip", B, skip.H, skip.W, skip.C, skip.ld)
            N.check(L.vs_upcat2x_bwd(N.ptr(dcat.t), dcat.ld, B, xin.H, xin.W, xin.C, skip.C, 2 ** -0.5, N.ptr(dxin.t), dxin.ld, N.ptr(dsk.t), dsk.ld, st),
                    "vs_upcat2x_bwd")
            dskips[nlev - k] = dsk                      # hid index of that skip: k = 0 pops hid[nlev] (= h3), k = nlev - 1 pops hid[1]
            dcur = dxin
        # ---- bottleneck
        for j in range(c.num_blocks - 1, -1, -1):
            add = dskips.pop(nlev) if j == 0 else None                          # h3 also feeds ups[0] as its skip
            dcur = self._resblock_bwd(eng, S["bott"][j], E["bott"][j], f"{u}.bottleneck.model.{j}", dcur, G, f"e.g.bott{j}", dx_add=add)
        # ---- message channels of h3 = [lat | msg]
        dh3 = dcur
        Bm = S["msgs"].shape[0]
        colm = eng.buf("tr.e.g.h3mean", B * dh3.ld)
        N.check(L.vs_colmean(N.ptr(dh3.t), B, dh3.H * dh3.W, dh3.ld, N.ptr(colm), st), "vs_colmean")
        dlat = (colm.view(B, dh3.ld)[:, c.zc[-1]: c.zc[-1] + c.hidden] * float(dh3.H * dh3.W)).contiguous()
        if Bm == 1:
            dlat = dlat.sum(0, keepdim=True).contiguous()
        dtab = torch.empty(2 * c.nbits, c.hidden, device=eng.dev, dtype=torch.float32)
        N.check(L.vs_msg_table_grad(N.ptr(dlat), N.ptr(S["msgs"]), Bm, c.nbits, c.hidden, N.ptr(dtab), st), "vs_msg_table_grad")
        G[u + ".msg_processor.msg_embeddings.weight"] = dtab
        dcur = Act(dh3.t, B, dh3.H, dh3.W, c.zc[-1], dh3.ld)                   # the latent part flows on into the last down block
        # ---- down path
        for i in range(nlev - 1, -1, -1):
            rec = S["downs"][i]
            name = f"{u}.downs.{i}"
            ddwn = self._resblock_bwd(eng, rec["rb"], E["downs"][i]["rb"], name + ".conv", dcur, G, f"e.g.down{i}")
            src, dwn = rec["src"], rec["dwn"]
            co, ci = dwn.C, src.C
            G[name + ".down.weight"] = self._conv3_wgrad(eng, ddwn, co, src, ci, stride=2)
            G[name + ".down.bias"] = h._colsum(eng, ddwn, co)
            dil = h._act(eng, f"e.g.down{i}.dil", B, src.H, src.W, co, ddwn.ld)
            N.check(L.vs_dilate2(N.ptr(ddwn.t), B, dwn.H, dwn.W, ddwn.ld, src.H, src.W, N.ptr(dil.t), st), "vs_dilate2")
            dsrc = h._act(eng, f"e.g.down{i}.dsrc", B, src.H, src.W, ci, src.ld)
            eng.conv(dil, self._flip_t(g(name + ".down.weight"), dil.ld), dsrc, pad=1, arith=BWD_ARITH, res=dskips.pop(i, None))
            dcur = dsrc
        self._resblock_bwd(eng, S["inc"], E["inc"], u + ".inc", dcur, G, "e.g.inc", need_dx=False)       # the frames need no gradient
        return G


class GeneratorStep:
    """One accumulation step of train.py:626-643 with the generator branch of `VideosealLoss` (videosealloss.py:111-192, optimizer_idx 0)
    computed on the HIP path end to end: differentiable forward (autograd.py) -> perceptual ('mse' / 'yuv') and decoding terms
    (vs_percep_mse, vs_bce_logits) -> adaptive weights through `get_last_layer()` (videosealloss.py:72-107: one backward probe per term that
    stops at the output convolution) -> backward into `.grad` of every embedder and detector parameter.  The discriminator term is a second
    trainable network outside this path (disc_weight = 0, what train.py itself uses for lambda_d = 0); the detection term needs a per-pixel
    mask head, which the ConvNeXt / ViT extractors of the shipped cards do not have (the reference's BCE raises on the shape mismatch too).

    The reference's own loss object works as well -- `model(imgs, masks)` returns graph-carrying tensors -- this class is the same step
    without ATen in the loss."""

    def __init__(self, model, percep_loss: str = "mse", percep_weight: float = 1.0, decode_weight: float = 0.0, detect_weight: float = 0.0,
                 balanced: bool = True, total_norm: float = 0.0, temperature: float = 1.0):
        if detect_weight > 0:
            raise NotImplementedError("detect_weight > 0: the per-frame extractors predict no mask map (videosealloss.py:140-147 needs [b,1,h,w] logits)")
        if percep_weight > 0 and percep_loss not in ("mse", "yuv"):
            raise NotImplementedError(f"perceptual loss {percep_loss!r}: 'mse' and 'yuv' run on the HIP path (the rest are pretrained networks)")
        self.model, self.percep_loss, self.temperature = model, percep_loss, float(temperature)
        self.percep_weight, self.decode_weight, self.balanced, self.total_norm = percep_weight, decode_weight, balanced, total_norm

    def losses(self, imgs: torch.Tensor, outputs: dict):
        from . import autograd as AG
        losses, weights = {}, {}
        dev = outputs["imgs_w"].device
        if self.percep_weight > 0:
            losses["percep"], weights["percep"] = AG.percep_loss(imgs.to(dev), outputs["imgs_w"], self.percep_loss), self.percep_weight
        if self.decode_weight > 0:
            losses["decode"], weights["decode"] = AG.decoding_loss(outputs["preds"], outputs["msgs"], self.temperature), self.decode_weight
        return losses, weights

    def scales(self, losses: dict, weights: dict):
        """videosealloss.py:72-107: (w_i / sum w) * N / (eps + ||d loss_i / d last_layer||), N = total_norm or the LAST term's gradient norm"""
        last = self.model.embedder.get_last_layer()
        if not (self.balanced and last.requires_grad):
            return dict(weights)
        norms = []
        for v in losses.values():
            g = torch.autograd.grad(v, last, retain_graph=True, allow_unused=True)[0] if v.requires_grad else None
            norms.append(torch.norm(g) if g is not None else torch.zeros((), device=last.device))
        tot = sum(weights.values())
        n = norms[-1] if self.total_norm <= 0 else self.total_norm
        return {k: (w / tot) * n / (1e-12 + gn) for (k, w), gn in zip(weights.items(), norms)}

    def step(self, imgs: torch.Tensor, masks: torch.Tensor, msgs: Optional[torch.Tensor] = None, is_video: bool = False,
             accumulation_steps: int = 1):
        """returns (total_loss, log, outputs); gradients are accumulated into `.grad` scaled by 1 / accumulation_steps (train.py:641)"""
        outputs = self.model(imgs, masks, msgs, is_video=is_video)
        losses, weights = self.losses(imgs, outputs)
        scales = self.scales(losses, weights)
        total = sum(scales[k] * losses[k] for k in losses)
        (total / accumulation_steps).backward()
        log = {"total_loss": total.detach(), **{f"loss_{k}": v.detach() for k, v in losses.items()},
               **{f"scale_{k}": torch.as_tensor(v).detach() for k, v in scales.items()}}
        return total.detach(), log, outputs
