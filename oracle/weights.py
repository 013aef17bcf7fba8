"""Model spec + seeded synthetic state_dict (TEST INFRASTRUCTURE, see oracle/__init__.py).

No checkpoint is available offline, so parity is "same seeded state_dict loaded
into the reference, the oracle and the HIP candidate".  The generator below
enumerates the reference's state_dict layout by itself (names/shapes follow
videoseal/modules/unet.py:123-166, modules/convnext.py:100-134,
modules/pixel_decoder.py:42-57, modules/jnd.py:26-58, data/transforms.py:15-21)
and fills every tensor with *non-trivial* values: BatchNorm running stats,
LayerNorm affine and GRN gamma/beta are randomised (their defaults are
identity/zero and would hide bugs).  ``tests/golden/make_golden.py`` checks the
key/shape list against the real reference modules (strict load).
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass, field
from typing import Dict, List

import torch
import yaml


@dataclass
class ModelSpec:
    nbits: int = 256
    hidden: int = 256                 # embedder.py:244  int(nbits * multiplier)
    img_size: int = 256
    scaling_w: float = 0.2
    scaling_i: float = 1.0
    chunk_size: int = 32
    step_size: int = 4
    yuv: bool = True                  # embedder.py:281  'yuv' in model name
    # U-Net (unet.py:110-166)
    in_ch: int = 1
    out_ch: int = 1
    z: int = 16
    mults: List[int] = field(default_factory=lambda: [1, 2, 4, 8])
    num_blocks: int = 8
    last_tanh: bool = True
    # ConvNeXt-V2 extractor (convnext.py:89-134, extractor.py:189-208)
    depths: List[int] = field(default_factory=lambda: [3, 3, 9, 3])
    dims: List[int] = field(default_factory=lambda: [96, 192, 384, 768])
    stem_stride: int = 4
    # JND (attenuation.yaml jnd_1_1); jnd_in = 0: the card has no attenuation (videoseal_0.0: `attenuation: None`, cfg.py:126-131)
    jnd_in: int = 1
    jnd_out: int = 1
    # U-Net flavour (common.py:110-127, 182-194): 'relu' + 'batch' (released 1.0 / PixelSeal / ChunkySeal) or 'silu' + 'rms' (legacy 0.0)
    unet_act: str = "relu"
    unet_norm: str = "batch"
    # extractor family (extractor.py:170-213): 'convnext' or 'sam' (ImageEncoderViT, vit.py:14-144)
    extractor: str = "convnext"
    vit_dim: int = 384
    vit_depth: int = 12
    vit_heads: int = 6
    vit_patch: int = 16
    vit_window: int = 8
    vit_global: List[int] = field(default_factory=lambda: [2, 5, 8, 11])
    vit_out: int = 384
    vit_mlp_ratio: float = 4.0
    vit_rel_pos: bool = True

    @property
    def zc(self) -> List[int]:
        return [self.z * m for m in self.mults]

    @property
    def bott(self) -> int:
        return self.zc[-1] + self.hidden


def spec_from_card(path: str) -> ModelSpec:
    """Card YAML -> ModelSpec, following cfg.py:88-122 and the two builders."""
    card = yaml.safe_load(open(path))
    a = card["args"]
    u = card["embedder"]["params"]["unet"]
    e = card["extractor"]["params"]
    nbits = int(a["nbits"])
    mult = a.get("hidden_size_multiplier", 2)
    dims = list(e["encoder"].get("dims", [0, 0, 0, 0]))
    if e.get("proportional_dim", False):          # extractor.py:193-198
        m = math.sqrt(nbits / 128)
        dims = [int(d * m) for d in dims]
    att = {"jnd_1_1": (1, 1), "jnd_3_3": (3, 3), "jnd_1_3": (1, 3), "jnd_3_1": (3, 1)}.get(str(a["attenuation"]).lower(), (0, 0))
    extra = {}
    if str(card["extractor"]["model"]).startswith("sam"):
        v = e["encoder"]
        extra = dict(extractor="sam", vit_dim=int(v["embed_dim"]), vit_depth=int(v["depth"]), vit_heads=int(v["num_heads"]),
                     vit_patch=int(v["patch_size"]), vit_window=int(v["window_size"]), vit_global=list(v["global_attn_indexes"]),
                     vit_out=int(v["out_chans"]), vit_mlp_ratio=float(v["mlp_ratio"]), vit_rel_pos=bool(v["use_rel_pos"]))
        dims = [0, 0, 0, int(v["out_chans"])]
    return ModelSpec(
        unet_act=str(u.get("activation", "relu")), unet_norm=("rms" if str(u.get("normalization", "batch")).startswith("rms") else "batch"),
        **extra,
        nbits=nbits, hidden=int(nbits * mult), img_size=int(a.get("img_size_proc", a.get("img_size_extractor", 256))),
        scaling_w=float(a["scaling_w"]), scaling_i=float(a["scaling_i"]),
        chunk_size=int(a.get("videoseal_chunk_size", a.get("videowam_chunk_size"))),
        step_size=int(a.get("videoseal_step_size", a.get("videowam_step_size"))),
        yuv="yuv" in card["embedder"]["model"],
        in_ch=int(u["in_channels"]), out_ch=int(u["out_channels"]), z=int(u["z_channels"]),
        mults=list(u["z_channels_mults"]), num_blocks=int(u["num_blocks"]),
        last_tanh=bool(u.get("last_tanh", True)),
        depths=list(e["encoder"].get("depths", [0, 0, 0, 0])), dims=dims,
        stem_stride=int(e["encoder"].get("stem_stride", 4)),
        jnd_in=att[0], jnd_out=att[1],
    )


def tiny_spec(**kw) -> ModelSpec:
    """Small architecture of the same family for fast CPU tests (the reference
    builders accept arbitrary sizes, so goldens exist for it too)."""
    d = dict(nbits=16, hidden=16, img_size=64, chunk_size=4, step_size=2, z=8, mults=[1, 2, 4, 8],
             num_blocks=2, depths=[1, 1, 2, 1], dims=[16, 32, 48, 64], stem_stride=4)
    d.update(kw)
    return ModelSpec(**d)


def legacy_tiny_spec(**kw) -> ModelSpec:
    """Small architecture of the videoseal_0.0 family: RMSNorm/SiLU RGB U-Net, ViT extractor on an 8 x 8 token grid with
    4 x 4 windows and two global blocks, no JND."""
    d = dict(nbits=16, hidden=32, img_size=64, chunk_size=4, step_size=2, z=8, mults=[1, 2, 4, 8], num_blocks=2, yuv=False,
             in_ch=3, out_ch=3, unet_act="silu", unet_norm="rms", extractor="sam", vit_dim=32, vit_depth=4, vit_heads=2,
             vit_patch=8, vit_window=4, vit_global=[1, 3], vit_out=24, depths=[0, 0, 0, 0], dims=[0, 0, 0, 24],
             jnd_in=0, jnd_out=0, scaling_w=1.0)
    d.update(kw)
    return ModelSpec(**d)


def state_dict_layout(s: ModelSpec) -> Dict[str, tuple]:
    """name -> shape for every tensor of Videoseal.state_dict()."""
    L: Dict[str, tuple] = {}

    def bn(p, c):
        L[p + ".weight"] = (c,); L[p + ".bias"] = (c,)
        L[p + ".running_mean"] = (c,); L[p + ".running_var"] = (c,)
        L[p + ".num_batches_tracked"] = ()

    def norm(p, c):                                  # common.py:182-194: BatchNorm2d or ChanRMSNorm (gamma [C,1,1])
        if s.unet_norm == "rms":
            L[p + ".gamma"] = (c, 1, 1)
        else:
            bn(p, c)

    def resblock(p, cin, cout):                      # unet.py:20-36
        L[p + ".double_conv.0.weight"] = (cout, cin, 3, 3); norm(p + ".double_conv.1", cout)
        L[p + ".double_conv.3.weight"] = (cout, cout, 3, 3); norm(p + ".double_conv.4", cout)
        L[p + ".res_conv.weight"] = (cout, cin, 1, 1); L[p + ".res_conv.bias"] = (cout,)

    zc = s.zc
    emb = (2 * s.nbits, s.hidden)
    L["embedder.unet.msg_processor.msg_embeddings.weight"] = emb
    u = "embedder.unet"
    resblock(u + ".inc", s.in_ch, zc[0])
    for i in range(len(zc) - 1):                     # unet.py:150-153, 74-78
        L[f"{u}.downs.{i}.down.weight"] = (zc[i + 1], zc[i], 3, 3); L[f"{u}.downs.{i}.down.bias"] = (zc[i + 1],)
        resblock(f"{u}.downs.{i}.conv", zc[i + 1], zc[i + 1])
    for j in range(s.num_blocks):
        resblock(f"{u}.bottleneck.model.{j}", s.bott, s.bott)
    zz = zc[:-1] + [s.bott]
    for k, i in enumerate(reversed(range(len(zz) - 1))):   # unet.py:160-163, 58-65
        cin, cout = 2 * zz[i + 1], zz[i]
        L[f"{u}.ups.{k}.up.upsample_block.2.weight"] = (cout, cin, 3, 3)
        L[f"{u}.ups.{k}.up.upsample_block.3.weight"] = (cout,); L[f"{u}.ups.{k}.up.upsample_block.3.bias"] = (cout,)
        resblock(f"{u}.ups.{k}.conv", cout, cout)
    L[u + ".outc.weight"] = (s.out_ch, zc[0], 1, 1); L[u + ".outc.bias"] = (s.out_ch,)
    L["embedder.msg_processor.msg_embeddings.weight"] = emb      # same tensor, registered twice (embedder.py:141-142)

    if s.extractor == "sam":                         # vit.py:55-127, 302-339; extractor.py:171-177
        ie = "detector.image_encoder"
        D_, g = s.vit_dim, s.img_size // s.vit_patch
        hd = D_ // s.vit_heads
        L[ie + ".pos_embed"] = (1, g, g, D_)
        L[ie + ".patch_embed.proj.weight"] = (D_, 3, s.vit_patch, s.vit_patch); L[ie + ".patch_embed.proj.bias"] = (D_,)
        for i in range(s.vit_depth):
            p = f"{ie}.blocks.{i}"
            t = g if (i in s.vit_global or s.vit_window == 0) else s.vit_window      # vit.py:89-90, 185
            L[p + ".norm1.weight"] = (D_,); L[p + ".norm1.bias"] = (D_,)
            if s.vit_rel_pos:
                L[p + ".attn.rel_pos_h"] = (2 * t - 1, hd); L[p + ".attn.rel_pos_w"] = (2 * t - 1, hd)
            L[p + ".attn.qkv.weight"] = (3 * D_, D_); L[p + ".attn.qkv.bias"] = (3 * D_,)
            L[p + ".attn.proj.weight"] = (D_, D_); L[p + ".attn.proj.bias"] = (D_,)
            L[p + ".norm2.weight"] = (D_,); L[p + ".norm2.bias"] = (D_,)
            hid = int(D_ * s.vit_mlp_ratio)
            L[p + ".mlp.lin1.weight"] = (hid, D_); L[p + ".mlp.lin1.bias"] = (hid,)
            L[p + ".mlp.lin2.weight"] = (D_, hid); L[p + ".mlp.lin2.bias"] = (D_,)
        O_ = s.vit_out
        L[ie + ".neck.0.weight"] = (O_, D_, 1, 1)
        L[ie + ".neck.1.weight"] = (O_,); L[ie + ".neck.1.bias"] = (O_,)
        L[ie + ".neck.2.weight"] = (O_, O_, 3, 3)
        L[ie + ".neck.3.weight"] = (O_,); L[ie + ".neck.3.bias"] = (O_,)
    c = "detector.convnext"
    d = s.dims if s.extractor == "convnext" else [0, 0, 0, s.vit_out]
    if s.extractor == "convnext":
        L[f"{c}.downsample_layers.0.0.weight"] = (d[0], 3, 4, 4); L[f"{c}.downsample_layers.0.0.bias"] = (d[0],)
        L[f"{c}.downsample_layers.0.1.weight"] = (d[0],); L[f"{c}.downsample_layers.0.1.bias"] = (d[0],)
        for i in range(3):
            L[f"{c}.downsample_layers.{i+1}.0.weight"] = (d[i],); L[f"{c}.downsample_layers.{i+1}.0.bias"] = (d[i],)
            L[f"{c}.downsample_layers.{i+1}.1.weight"] = (d[i + 1], d[i], 2, 2); L[f"{c}.downsample_layers.{i+1}.1.bias"] = (d[i + 1],)
        for st in range(4):
            for j in range(s.depths[st]):
                p = f"{c}.stages.{st}.{j}"; C = d[st]
                L[p + ".dwconv.weight"] = (C, 1, 7, 7); L[p + ".dwconv.bias"] = (C,)
                L[p + ".norm.weight"] = (C,); L[p + ".norm.bias"] = (C,)
                L[p + ".pwconv1.weight"] = (4 * C, C); L[p + ".pwconv1.bias"] = (4 * C,)
                L[p + ".grn.gamma"] = (1, 1, 1, 4 * C); L[p + ".grn.beta"] = (1, 1, 1, 4 * C)
                L[p + ".pwconv2.weight"] = (C, 4 * C); L[p + ".pwconv2.bias"] = (C,)
    pd = "detector.pixel_decoder"
    E = d[-1]
    L[pd + ".output_upscaling.0.upsample_block.2.weight"] = (E, E, 3, 3)
    L[pd + ".output_upscaling.0.upsample_block.3.weight"] = (E,); L[pd + ".output_upscaling.0.upsample_block.3.bias"] = (E,)
    L[pd + ".linear.weight"] = (s.nbits + 1, E); L[pd + ".linear.bias"] = (s.nbits + 1,)
    L["rgb2yuv.M"] = (3, 3)
    g = s.jnd_in
    if g > 0:                                        # no JND module in the card (cfg.py:131) -> no attenuation.* tensors
        L["attenuation.conv_x.weight"] = (g, 1, 3, 3); L["attenuation.conv_y.weight"] = (g, 1, 3, 3)
        L["attenuation.conv_lum.weight"] = (g, 1, 5, 5)
    return L


_SOBEL_X = [[-1., 0., 1.], [-2., 0., 2.], [-1., 0., 1.]]
_SOBEL_Y = [[1., 2., 1.], [0., 0., 0.], [-1., -2., -1.]]
_LUM = [[1., 1., 1., 1., 1.], [1., 2., 2., 2., 1.], [1., 2., 0., 2., 1.], [1., 2., 2., 2., 1.], [1., 1., 1., 1., 1.]]
_YUV = [[0.299, 0.587, 0.114], [-0.14713, -0.28886, 0.436], [0.615, -0.51499, -0.10001]]


def make_state_dict(s: ModelSpec, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic weights.  Each tensor gets its own generator seeded from
    (seed, crc32(name)), so the values do not depend on enumeration order."""
    L = state_dict_layout(s)
    sd: Dict[str, torch.Tensor] = {}

    def rnd(name, shape, kind):
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
        if kind == "normal":
            return torch.randn(shape, generator=g)
        return torch.rand(shape, generator=g)

    for name, shape in L.items():
        if name == "embedder.msg_processor.msg_embeddings.weight":
            continue
        leaf = name.rsplit(".", 1)[-1]
        if name == "rgb2yuv.M":
            t = torch.tensor(_YUV)
        elif name.startswith("attenuation."):
            k = {"conv_x": _SOBEL_X, "conv_y": _SOBEL_Y, "conv_lum": _LUM}[name.split(".")[1]]
            t = torch.tensor(k)[None, None].repeat(shape[0], 1, 1, 1)
        elif leaf == "num_batches_tracked":
            t = torch.tensor(100, dtype=torch.int64)
        elif leaf == "running_mean":
            t = 0.1 * rnd(name, shape, "normal")
        elif leaf == "running_var":
            t = 0.5 + rnd(name, shape, "uniform")
        elif leaf == "gamma" and len(shape) == 3:    # ChanRMSNorm scale (default ones)
            t = 0.5 + rnd(name, shape, "uniform")
        elif leaf == "pos_embed":
            t = 0.2 * rnd(name, shape, "normal")
        elif leaf in ("rel_pos_h", "rel_pos_w"):     # zero-initialised in the reference: randomised here or the term would go untested
            t = 0.3 * rnd(name, shape, "normal")
        elif leaf == "gamma":
            t = 0.5 * rnd(name, shape, "normal")
        elif leaf == "beta":
            t = 0.1 * rnd(name, shape, "normal")
        elif "msg_embeddings" in name:
            t = rnd(name, shape, "normal") / math.sqrt(s.nbits)
        elif len(shape) == 1:                        # norm affine or conv/linear bias
            is_norm_w = leaf == "weight"
            t = (0.5 + rnd(name, shape, "uniform")) if is_norm_w else 0.1 * rnd(name, shape, "normal")
        else:                                        # conv / linear weight
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            gain = math.sqrt(2.0)
            if ".outc." in name:
                gain = 0.25
            elif "res_conv" in name or "pwconv2" in name or ".linear." in name or ".attn." in name or ".mlp.lin2" in name or ".neck." in name:
                gain = 1.0
            t = rnd(name, shape, "normal") * (gain / math.sqrt(fan_in))
        sd[name] = t.to(torch.int64 if leaf == "num_batches_tracked" else torch.float32).contiguous()
    sd["embedder.msg_processor.msg_embeddings.weight"] = sd["embedder.unet.msg_processor.msg_embeddings.weight"]
    return sd
