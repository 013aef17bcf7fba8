"""Functional fp32 CPU restatement of the VideoSeal embed / detect path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function takes the
reference-format state_dict ``sd`` (NCHW fp32 tensors) and plain tensors; the
primitives are torch ATen CPU ops, which are what the reference itself runs.
Pinned against the real reference by tests/golden (make_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .weights import ModelSpec

SD = Dict[str, torch.Tensor]
AA = {"mode": "bilinear", "align_corners": False, "antialias": True}


# --------------------------------------------------------------------------- layers
def _bn(sd: SD, p: str, x, bn: Optional[dict] = None):
    """nn.BatchNorm2d (unet.py:26,30 via common.py:182-184), eps = 1e-5.  bn=None: eval mode (running statistics).
    bn = {}: TRAIN mode -- batch statistics, and the updated running_mean / running_var / num_batches_tracked
    (momentum 0.1, unbiased variance) are written into `bn` under their state_dict names (sd itself is not modified)."""
    if bn is None:
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                            training=False, eps=1e-5)
    rm, rv = sd[p + ".running_mean"].clone(), sd[p + ".running_var"].clone()
    y = F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], training=True, momentum=0.1, eps=1e-5)
    bn[p + ".running_mean"], bn[p + ".running_var"] = rm, rv
    bn[p + ".num_batches_tracked"] = sd[p + ".num_batches_tracked"] + 1
    return y


def _chan_rms(sd: SD, p: str, x):
    """common.py:172-179  ChanRMSNorm: F.normalize(x, dim=1) * sqrt(C) * gamma   (F.normalize: x / max(||x||_2, 1e-12))."""
    return F.normalize(x, dim=1) * (x.shape[1] ** 0.5) * sd[p + ".gamma"]


def resnet_block(sd: SD, p: str, x, bn: Optional[dict] = None):
    """unet.py:17-39  act(norm(conv3(act(norm(conv3(x)))))) + conv1x1(x); norm/act = BatchNorm/ReLU (released cards) or
    ChanRMSNorm/SiLU (legacy videoseal_0.0 card) -- told apart by the tensors present."""
    if p + ".double_conv.1.gamma" in sd:
        h = F.silu(_chan_rms(sd, p + ".double_conv.1", F.conv2d(x, sd[p + ".double_conv.0.weight"], padding=1)))
        h = F.silu(_chan_rms(sd, p + ".double_conv.4", F.conv2d(h, sd[p + ".double_conv.3.weight"], padding=1)))
        return h + F.conv2d(x, sd[p + ".res_conv.weight"], sd[p + ".res_conv.bias"])
    h = F.relu(_bn(sd, p + ".double_conv.1", F.conv2d(x, sd[p + ".double_conv.0.weight"], padding=1), bn))
    h = F.relu(_bn(sd, p + ".double_conv.4", F.conv2d(h, sd[p + ".double_conv.3.weight"], padding=1), bn))
    return h + F.conv2d(x, sd[p + ".res_conv.weight"], sd[p + ".res_conv.bias"])


def layernorm_cf(x, w, b, eps=1e-6):
    """common.py:147-155  channels_first LayerNorm (biased variance over C)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


def upsample_block(sd: SD, p: str, x, factor: int, act):
    """common.py:45-52  bilinear xf (align_corners=False) -> ReflectionPad(1) -> conv3x3 (no bias) -> LN(cf) -> act."""
    if factor != 1:
        x = F.interpolate(x, scale_factor=factor, mode="bilinear", align_corners=False)
    else:
        # nn.Upsample(scale_factor=1, bilinear) is an exact identity in ATen
        x = F.interpolate(x, scale_factor=1, mode="bilinear", align_corners=False)
    x = F.pad(x, (1, 1, 1, 1), mode="reflect")
    x = F.conv2d(x, sd[p + ".upsample_block.2.weight"])
    x = layernorm_cf(x, sd[p + ".upsample_block.3.weight"], sd[p + ".upsample_block.3.bias"])
    return act(x)


def msg_latent(sd: SD, msgs):
    """msg_processor.py:88-98  idx = 2k + bit, embedding lookup, sum over k -> [b, hidden]."""
    table = sd["embedder.unet.msg_processor.msg_embeddings.weight"]
    k = msgs.shape[-1]
    idx = (2 * torch.arange(k)[None, :] + msgs.long())
    return F.embedding(idx, table).sum(dim=-2)


def unet_forward(sd: SD, s: ModelSpec, x, msgs, bn: Optional[dict] = None):
    """unet.py:170-197 (x already in [-1,1]).  bn: see _bn (None = eval, dict = train mode)."""
    u = "embedder.unet"
    hid = [resnet_block(sd, u + ".inc", x, bn)]
    for i in range(len(s.mults) - 1):
        d = F.conv2d(hid[-1], sd[f"{u}.downs.{i}.down.weight"], sd[f"{u}.downs.{i}.down.bias"], stride=2, padding=1)
        hid.append(resnet_block(sd, f"{u}.downs.{i}.conv", d, bn))
    lat = hid.pop()
    m = msg_latent(sd, msgs)[:, :, None, None].expand(-1, -1, lat.shape[-2], lat.shape[-1])
    hid.append(torch.cat([lat, m], dim=1))          # msg_processor.py:111-115 (concat, msg_mult = 1)
    x = hid[-1]
    for j in range(s.num_blocks):
        x = resnet_block(sd, f"{u}.bottleneck.model.{j}", x, bn)
    for k in range(len(s.mults) - 1):
        x = torch.cat((x, hid.pop() * (2 ** -0.5)), dim=1)     # unet.py:186-187
        x = upsample_block(sd, f"{u}.ups.{k}.up", x, 2, F.silu if s.unet_act == "silu" else F.relu)      # unet.py:61-62: the U-Net's act_layer
        x = resnet_block(sd, f"{u}.ups.{k}.conv", x, bn)
    x = F.conv2d(x, sd[u + ".outc.weight"], sd[u + ".outc.bias"])
    return torch.tanh(x) if s.last_tanh else x


def embedder_forward(sd: SD, s: ModelSpec, imgs01, msgs, bn: Optional[dict] = None):
    """embedder.py:151-165  x*2-1 -> UNetMsg."""
    return unet_forward(sd, s, imgs01 * 2 - 1, msgs, bn)


def convnext_block(sd: SD, p: str, x):
    """convnext.py:41-57 + common.py:158-169 (GRN)."""
    C = x.shape[1]
    h = F.conv2d(x, sd[p + ".dwconv.weight"], sd[p + ".dwconv.bias"], padding=3, groups=C)
    h = h.permute(0, 2, 3, 1)
    h = F.layer_norm(h, (C,), sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    h = F.gelu(F.linear(h, sd[p + ".pwconv1.weight"], sd[p + ".pwconv1.bias"]))
    gx = torch.norm(h, p=2, dim=(1, 2), keepdim=True)
    nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
    h = sd[p + ".grn.gamma"] * (h * nx) + sd[p + ".grn.beta"] + h
    h = F.linear(h, sd[p + ".pwconv2.weight"], sd[p + ".pwconv2.bias"])
    return x + h.permute(0, 3, 1, 2)


def _rel_pos(q_size: int, k_size: int, rel_pos):
    """vit.py:405-433 get_rel_pos for q_size == k_size and a table of 2*size-1 rows (the only case the encoder produces):
    R[i, j] = rel_pos[i - j + size - 1]."""
    assert q_size == k_size and rel_pos.shape[0] == 2 * q_size - 1
    idx = torch.arange(q_size)[:, None] - torch.arange(k_size)[None, :] + (k_size - 1)
    return rel_pos[idx]


def vit_attention(sd: SD, p: str, x, heads: int, rel_pos: bool):
    """vit.py:302-360 Attention.forward on [B, H, W, C] tokens (B = frames * windows), decomposed relative positions vit.py:436-470."""
    B, H, W, C = x.shape
    hd = C // heads
    qkv = F.linear(x, sd[p + ".qkv.weight"], sd[p + ".qkv.bias"]).reshape(B, H * W, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, B * heads, H * W, hd).unbind(0)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    if rel_pos:
        Rh, Rw = _rel_pos(H, H, sd[p + ".rel_pos_h"]), _rel_pos(W, W, sd[p + ".rel_pos_w"])
        rq = q.reshape(B * heads, H, W, hd)
        rel_h = torch.einsum("bhwc,hkc->bhwk", rq, Rh)
        rel_w = torch.einsum("bhwc,wkc->bhwk", rq, Rw)
        attn = (attn.view(B * heads, H, W, H, W) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(B * heads, H * W, H * W)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).view(B, heads, H, W, hd).permute(0, 2, 3, 1, 4).reshape(B, H, W, C)
    return F.linear(x, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


def vit_block(sd: SD, p: str, x, heads: int, window: int, rel_pos: bool):
    """vit.py:146-209 Block.forward: x + attn(norm1(x)) (windowed when window > 0, vit.py:363-402), then x + mlp(norm2(x))."""
    B, H, W, C = x.shape
    h = F.layer_norm(x, (C,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)      # nn.LayerNorm default eps
    if window > 0:
        ph, pw = (window - H % window) % window, (window - W % window) % window
        h = F.pad(h, (0, 0, 0, pw, 0, ph))
        Hp, Wp = H + ph, W + pw
        h = h.view(B, Hp // window, window, Wp // window, window, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, window, window, C)
    h = vit_attention(sd, p + ".attn", h, heads, rel_pos)
    if window > 0:
        h = h.view(B, Hp // window, Wp // window, window, window, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)[:, :H, :W]
    x = x + h
    h = F.layer_norm(x, (C,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    h = F.linear(F.gelu(F.linear(h, sd[p + ".mlp.lin1.weight"], sd[p + ".mlp.lin1.bias"])), sd[p + ".mlp.lin2.weight"], sd[p + ".mlp.lin2.bias"])
    return x + h


def vit_extractor_forward(sd: SD, s: ModelSpec, imgs01):
    """extractor.py:45-75 SegmentationExtractor.forward: x*2-1 -> ImageEncoderViT (vit.py:129-144) -> PixelDecoder."""
    ie = "detector.image_encoder"
    x = F.conv2d(imgs01 * 2 - 1, sd[ie + ".patch_embed.proj.weight"], sd[ie + ".patch_embed.proj.bias"], stride=s.vit_patch)
    x = x.permute(0, 2, 3, 1) + sd[ie + ".pos_embed"]
    for i in range(s.vit_depth):
        x = vit_block(sd, f"{ie}.blocks.{i}", x, s.vit_heads, 0 if i in s.vit_global else s.vit_window, s.vit_rel_pos)
    x = x.permute(0, 3, 1, 2)
    x = layernorm_cf(F.conv2d(x, sd[ie + ".neck.0.weight"]), sd[ie + ".neck.1.weight"], sd[ie + ".neck.1.bias"])
    x = layernorm_cf(F.conv2d(x, sd[ie + ".neck.2.weight"], padding=1), sd[ie + ".neck.3.weight"], sd[ie + ".neck.3.bias"])
    x = upsample_block(sd, "detector.pixel_decoder.output_upscaling.0", x, 1, F.gelu)
    x = x.mean(dim=[-2, -1])
    return F.linear(x, sd["detector.pixel_decoder.linear.weight"], sd["detector.pixel_decoder.linear.bias"])


def extractor_forward(sd: SD, s: ModelSpec, imgs01):
    """extractor.py:154-167, convnext.py:146-156, pixel_decoder.py:61-83."""
    if s.extractor == "sam":
        return vit_extractor_forward(sd, s, imgs01)
    c = "detector.convnext"
    x = imgs01 * 2 - 1
    x = F.conv2d(x, sd[f"{c}.downsample_layers.0.0.weight"], sd[f"{c}.downsample_layers.0.0.bias"], stride=s.stem_stride)
    x = layernorm_cf(x, sd[f"{c}.downsample_layers.0.1.weight"], sd[f"{c}.downsample_layers.0.1.bias"])
    for st in range(4):
        if st > 0:
            x = layernorm_cf(x, sd[f"{c}.downsample_layers.{st}.0.weight"], sd[f"{c}.downsample_layers.{st}.0.bias"])
            x = F.conv2d(x, sd[f"{c}.downsample_layers.{st}.1.weight"], sd[f"{c}.downsample_layers.{st}.1.bias"], stride=2)
        for j in range(s.depths[st]):
            x = convnext_block(sd, f"{c}.stages.{st}.{j}", x)
    x = upsample_block(sd, "detector.pixel_decoder.output_upscaling.0", x, 1, F.gelu)
    x = x.mean(dim=[-2, -1])
    return F.linear(x, sd["detector.pixel_decoder.linear.weight"], sd["detector.pixel_decoder.linear.bias"])


# --------------------------------------------------------------------------- JND
def jnd_heatmaps(sd: SD, s: ModelSpec, imgs, clc: float = 0.3):
    """jnd.py:63-108 (in_channels=1,out_channels=1 is the only mode the cards use)."""
    x = 255 * imgs
    if s.jnd_in == 1:
        x = 0.299 * x[..., 0:1, :, :] + 0.587 * x[..., 1:2, :, :] + 0.114 * x[..., 2:3, :, :]
    g = s.jnd_in
    la = F.conv2d(x, sd["attenuation.conv_lum.weight"], padding=2, groups=g) / 32
    low = la <= 127
    la = torch.where(low, 17 * (1 - torch.sqrt(la / 127 + 1e-5)), 3 / 128 * (la - 127) + 3)
    gx = F.conv2d(x, sd["attenuation.conv_x.weight"], padding=1, groups=g)
    gy = F.conv2d(x, sd["attenuation.conv_y.weight"], padding=1, groups=g)
    cm = torch.sqrt(gx ** 2 + gy ** 2)
    cm = 0.117 * (16 * cm ** 2.4 / (cm ** 2 + 26 ** 2))
    h = torch.clamp_min(la + cm - clc * torch.minimum(la, cm), 0)
    if s.jnd_out == 3 and s.jnd_in == 1:
        h = h.repeat(1, 3, 1, 1)
    elif s.jnd_out == 1 and s.jnd_in == 3:
        h = torch.sum(h / 3, dim=1, keepdim=True)
    return h / 255


def rgb2y(sd: SD, x):
    """data/transforms.py:23-27 (full matmul, row 0 consumed by wam.py:168-170)."""
    yuv = torch.matmul(x.permute(0, 2, 3, 1).contiguous(), sd["rgb2yuv.M"].T).permute(0, 3, 1, 2).contiguous()
    return yuv[:, 0:1]


def _resize(x, size, interp):
    return F.interpolate(x, size=size, **interp) if tuple(x.shape[-2:]) != tuple(size) else x.clone()


# --------------------------------------------------------------------------- embed / detect
def embed_image(sd: SD, s: ModelSpec, imgs, msgs, interp=AA, lowres_attenuation=False, attenuate=True, clamp=True):
    """wam.py:134-204."""
    attenuate = attenuate and s.jnd_in > 0          # cards without a JND module (cfg.py:126-131)
    P = (s.img_size, s.img_size)
    res = _resize(imgs, P, interp)
    x = rgb2y(sd, res) if s.yuv else res
    preds = embedder_forward(sd, s, x, msgs)
    if attenuate and lowres_attenuation:
        preds = jnd_heatmaps(sd, s, res) * preds
    if tuple(imgs.shape[-2:]) != P:
        preds = F.interpolate(preds, size=imgs.shape[-2:], **interp)
    if attenuate and not lowres_attenuation:
        preds = jnd_heatmaps(sd, s, imgs) * preds
    out = s.scaling_i * imgs + s.scaling_w * preds
    if clamp:
        out = torch.clamp(out, 0, 1)
    return {"imgs_w": out, "preds_w": preds, "msgs": msgs}


def apply_video_mode(preds, total, step, mode):
    """videoseal.py:80-118."""
    if mode == "repeat":
        preds = torch.repeat_interleave(preds, step, dim=0)
    elif mode == "alternate":
        full = torch.zeros((total,) + preds.shape[1:])
        full[::step] = preds
        preds = full
    elif mode == "interpolate":
        full = torch.zeros((total,) + preds.shape[1:])
        alpha = 1 - torch.linspace(0, 1, steps=step)
        alpha = alpha.repeat((total - 1) // step).view(-1, 1, 1, 1)
        a = torch.repeat_interleave(preds[:-1], step, dim=0)
        b = torch.repeat_interleave(preds[1:], step, dim=0)
        inter = alpha * a + (1 - alpha) * b
        full[:len(inter)] = inter
        full[len(inter):] = preds[-1]
        preds = full
    return preds[:total]


def embed_video(sd: SD, s: ModelSpec, imgs, msgs, interp=AA, lowres_attenuation=False,
                chunk_size: Optional[int] = None, step_size: Optional[int] = None, video_mode="repeat",
                attenuate=True, clamp=True):
    """videoseal.py:258-350 (msgs is [1,k])."""
    attenuate = attenuate and s.jnd_in > 0          # cards without a JND module (cfg.py:126-131)
    assert msgs.shape[0] == 1, "Message should be unique"
    ck = chunk_size or s.chunk_size
    st = step_size or s.step_size
    P = (s.img_size, s.img_size)
    m = msgs.repeat(ck, 1)
    out = torch.zeros_like(imgs)
    nkey = len(imgs[::st])
    for ii in range(0, nkey, ck):
        n = min(ck, nkey - ii)
        a, b = ii * st, ii * st + n * st
        fr = imgs[a:b]
        if n < ck:
            m = m[:n]
        res = _resize(fr, P, interp)
        key = res[::st]
        x = rgb2y(sd, key) if s.yuv else key
        preds = apply_video_mode(embedder_forward(sd, s, x, m), len(fr), st, video_mode)
        if attenuate and lowres_attenuation:
            preds = jnd_heatmaps(sd, s, res) * preds
        if tuple(fr.shape[-2:]) != P:
            preds = F.interpolate(preds, size=fr.shape[-2:], **interp)
        if attenuate and not lowres_attenuation:
            preds = jnd_heatmaps(sd, s, fr) * preds
        out[a:b] = s.scaling_i * fr + s.scaling_w * preds
    if clamp:
        out = torch.clamp(out, 0, 1)
    return {"imgs_w": out, "msgs": msgs[0:1].repeat(len(imgs), 1)}


def forward_image(sd: SD, s: ModelSpec, imgs, masks, msgs, augment, interp=AA, bn: Optional[dict] = None, attenuate=True,
                  clamp=True, scaling_i=None, scaling_w=None):
    """wam.py:68-132, the training forward.  `augment(imgs_w, imgs, masks, is_video, do_resize) -> (imgs_aug, masks, name)` is the
    augmenter (oracle/augment.py:Augmenter).  bn: see _bn."""
    attenuate = attenuate and s.jnd_in > 0          # cards without a JND module (cfg.py:126-131)
    si = s.scaling_i if scaling_i is None else scaling_i
    sw = s.scaling_w if scaling_w is None else scaling_w
    P = (s.img_size, s.img_size)
    res = _resize(imgs, P, interp)
    x = rgb2y(sd, res) if s.yuv else res
    preds_w = embedder_forward(sd, s, x, msgs, bn)
    if tuple(imgs.shape[-2:]) != P:
        preds_w = F.interpolate(preds_w, size=imgs.shape[-2:], **interp)
    imgs_w = si * imgs + sw * preds_w                               # blender.py:61-68
    if attenuate:
        imgs_w = imgs + jnd_heatmaps(sd, s, imgs) * (imgs_w - imgs)      # jnd.py:110-114
    if clamp:
        imgs_w = torch.clamp(imgs_w, 0, 1)
    imgs_aug, masks, selected = augment(imgs_w, imgs, masks, False, False)
    if tuple(imgs_aug.shape[-2:]) != P:
        imgs_aug = F.interpolate(imgs_aug, size=P, **interp)
    preds = extractor_forward(sd, s, imgs_aug)
    return {"msgs": msgs, "masks": masks, "preds_w": preds_w, "imgs_w": imgs_w, "imgs_aug": imgs_aug, "preds": preds,
            "selected_aug": selected}


def forward_video(sd: SD, s: ModelSpec, imgs, masks, msgs, augment, interp=AA, bn: Optional[dict] = None, step_size=None,
                  video_mode="repeat", lowres_attenuation=False, attenuate=True, clamp=True):
    """videoseal.py:163-256 (`video_forward`)."""
    attenuate = attenuate and s.jnd_in > 0          # cards without a JND module (cfg.py:126-131)
    assert msgs.shape[0] == 1, "Message should be unique"
    st = step_size or s.step_size
    P = (s.img_size, s.img_size)
    m = msgs.expand(len(imgs), -1)
    res = _resize(imgs, P, interp)
    key = res[::st]
    x = rgb2y(sd, key) if s.yuv else key
    preds_w = apply_video_mode(embedder_forward(sd, s, x, m[::st], bn), len(res), st, video_mode)
    if lowres_attenuation and attenuate:
        preds_w = jnd_heatmaps(sd, s, res) * preds_w
        if tuple(imgs.shape[-2:]) != P:
            preds_w = F.interpolate(preds_w, size=imgs.shape[-2:], **interp)
        imgs_w = s.scaling_i * imgs + s.scaling_w * preds_w
    else:
        if tuple(imgs.shape[-2:]) != P:
            preds_w = F.interpolate(preds_w, size=imgs.shape[-2:], **interp)
        imgs_w = s.scaling_i * imgs + s.scaling_w * preds_w
        if attenuate:
            imgs_w = imgs + jnd_heatmaps(sd, s, imgs) * (imgs_w - imgs)
    if clamp:
        imgs_w = torch.clamp(imgs_w, 0, 1)
    imgs_aug, masks, selected = augment(imgs_w, imgs, masks, True, False)
    if tuple(imgs.shape[-2:]) != P:
        imgs_aug = F.interpolate(imgs_aug, size=P, **interp)
    preds = extractor_forward(sd, s, imgs_aug)
    return {"msgs": m, "masks": masks, "imgs_w": imgs_w, "imgs_aug": imgs_aug, "preds": preds, "selected_aug": selected}


def detect(sd: SD, s: ModelSpec, imgs, interp=AA):
    """wam.py:206-234 / videoseal.py:352-388 (chunking does not change values)."""
    res = _resize(imgs, (s.img_size, s.img_size), interp)
    return {"preds": extractor_forward(sd, s, res)}


def aggregate(bit_preds, aggregation="avg"):
    """videoseal.py:411-428."""
    if aggregation is None:
        d = bit_preds
    elif aggregation == "avg":
        d = bit_preds.mean(dim=0)
    elif aggregation == "squared_avg":
        d = (bit_preds * bit_preds.abs()).mean(dim=0)
    elif aggregation == "l1norm_avg":
        d = (bit_preds * torch.norm(bit_preds, p=1, dim=1).unsqueeze(1)).mean(dim=0)
    elif aggregation == "l2norm_avg":
        d = (bit_preds * torch.norm(bit_preds, p=2, dim=1).unsqueeze(1)).mean(dim=0)
    else:
        raise ValueError(aggregation)
    return d


def extract_message(sd: SD, s: ModelSpec, imgs, aggregation="avg",
                    interp={"mode": "bilinear", "align_corners": False, "antialias": False}):
    """videoseal.py:390-428."""
    preds = detect(sd, s, imgs, interp)["preds"]
    return (aggregate(preds[:, 1:], aggregation) > 0).squeeze().unsqueeze(0)


# --------------------------------------------------------------------------- metrics
def psnr(x, y, is_video=False):
    """evals/metrics.py:22-36."""
    delta = (255 * (x - y)).reshape(-1, x.shape[-3], x.shape[-2], x.shape[-1])
    dims = (0, 1, 2, 3) if is_video else (1, 2, 3)
    return 20 * math.log10(255.0) - 10 * torch.log10(torch.mean(delta ** 2, dim=dims))


def bit_accuracy(preds, targets, threshold=0.0):
    """evals/metrics.py:150-178 (non-pixelwise branch)."""
    return ((preds > threshold) == (targets > 0.5)).float().mean(dim=-1)
