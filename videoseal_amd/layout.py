"""Architecture description (ModelCfg) and the parameter tree of a Videoseal model.

The reference checkpoint format is part of the drop-in contract: 434 tensors for
VideoSeal 1.0 with keys such as ``embedder.unet.bottleneck.model.3.double_conv.1.running_var``.
Instead of re-declaring the reference's layer classes, the key space is generated from the
architecture numbers and materialised as a tree of plain containers (``ParamTree``); all compute
lives in the HIP engine.  tests/test_host.py checks keys, order and shapes against the key list
dumped from the real reference (tests/golden/state_dict_keys.json).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Tuple

import torch
import torch.nn as nn
import yaml


@dataclass
class ModelCfg:
    nbits: int = 256
    hidden: int = 256
    img_size: int = 256
    scaling_w: float = 0.2
    scaling_i: float = 1.0
    chunk_size: int = 32
    step_size: int = 4
    blending_method: str = "additive"
    yuv: bool = True
    in_ch: int = 1
    out_ch: int = 1
    z: int = 16
    mults: List[int] = field(default_factory=lambda: [1, 2, 4, 8])
    num_blocks: int = 8
    last_tanh: bool = True
    depths: List[int] = field(default_factory=lambda: [3, 3, 9, 3])
    dims: List[int] = field(default_factory=lambda: [96, 192, 384, 768])
    stem_stride: int = 4
    jnd_in: int = 1              # 0: the card has no JND attenuation (videoseal_0.0: `attenuation: None`, cfg.py:126-131)
    jnd_out: int = 1
    checkpoint_path: str = ""
    unet_act: str = "relu"       # 'relu' + 'batch' (1.0 / PixelSeal / ChunkySeal) or 'silu' + 'rms' (legacy 0.0 card), common.py:110-127, 182-194
    unet_norm: str = "batch"
    extractor: str = "convnext"  # or 'sam': ImageEncoderViT (vit.py:14-144) behind the same PixelDecoder
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


_JND_MODES = {"jnd_1_1": (1, 1), "jnd_3_3": (3, 3), "jnd_1_3": (1, 3), "jnd_3_1": (3, 1)}


def cfg_from_card(card: dict) -> ModelCfg:
    """Model-card dict (reference schema, cards/*.yaml) -> ModelCfg.
    Mirrors the decisions of utils/cfg.py:100-122, embedder.py:243-262,281 and extractor.py:189-203."""
    a = card["args"]
    emb, ext = card["embedder"], card["extractor"]
    if not str(emb["model"]).startswith("unet"):
        raise NotImplementedError(f"embedder '{emb['model']}': only the U-Net embedders of the released cards are built")
    sam = str(ext["model"]).startswith("sam")
    if not (sam or str(ext["model"]).startswith("convnext")):
        raise NotImplementedError(f"extractor '{ext['model']}': the ConvNeXt-V2 extractors of the released cards and the SAM-style ViT "
                                  f"of the legacy card are built")
    u, e = emb["params"]["unet"], ext["params"]
    mp = emb["params"].get("msg_processor", {})
    if mp.get("msg_processor_type", "binary+concat") != "binary+concat":
        raise NotImplementedError("only msg_processor_type 'binary+concat' is supported")
    act_, norm_ = str(u.get("activation", "relu")), str(u.get("normalization", "batch"))
    norm_ = "batch" if norm_.startswith("batch") else ("rms" if norm_.startswith("rms") else norm_)
    if (act_, norm_) not in (("relu", "batch"), ("silu", "rms")):
        raise NotImplementedError(f"U-Net activation/normalization '{act_}'/'{norm_}': relu + batch (released cards) or silu + rms (legacy card)")
    if list(e["pixel_decoder"].get("upscale_stages", [1])) != [1] or e["pixel_decoder"].get("pixelwise", False):
        raise NotImplementedError("pixel decoder: only upscale_stages [1], pixelwise False")
    if str(e["pixel_decoder"].get("upscale_type", "bilinear")) != "bilinear":
        raise NotImplementedError("pixel decoder: only upscale_type 'bilinear'")
    nbits = int(a["nbits"])
    mult = a.get("hidden_size_multiplier", 2)
    vit = {}
    if sam:
        v = e["encoder"]
        if not v.get("qkv_bias", True) or v.get("temporal_attention", False) or not v.get("use_abs_pos", True):
            raise NotImplementedError("ViT extractor: qkv_bias and absolute position embeddings are required, temporal attention is not built")
        vit = dict(extractor="sam", vit_dim=int(v["embed_dim"]), vit_depth=int(v["depth"]), vit_heads=int(v["num_heads"]),
                   vit_patch=int(v["patch_size"]), vit_window=int(v.get("window_size", 0)), vit_global=[int(i) for i in v.get("global_attn_indexes", [])],
                   vit_out=int(v["out_chans"]), vit_mlp_ratio=float(v.get("mlp_ratio", 4.0)), vit_rel_pos=bool(v.get("use_rel_pos", False)))
        dims, depths = [0, 0, 0, int(v["out_chans"])], [0, 0, 0, 0]
    else:
        dims = [int(v) for v in e["encoder"]["dims"]]
        depths = [int(v) for v in e["encoder"]["depths"]]
        if e.get("proportional_dim", False):
            f = math.sqrt(nbits / 128)
            dims = [int(v * f) for v in dims]
    att = str(a.get("attenuation", "jnd_1_1")).lower()
    if att.startswith("jnd") and att not in _JND_MODES:       # cfg.py:126-131: anything not starting with 'jnd' means no attenuation
        raise NotImplementedError(f"attenuation '{att}'")
    jnd = _JND_MODES[att] if att.startswith("jnd") else (0, 0)
    ck = a.get("videoseal_chunk_size", a.get("videowam_chunk_size", 8))
    stp = a.get("videoseal_step_size", a.get("videowam_step_size", 4))
    return ModelCfg(
        nbits=nbits, hidden=int(nbits * mult), img_size=int(a.get("img_size_proc", a.get("img_size_extractor", 256))),
        scaling_w=float(a.get("scaling_w", 1.0)), scaling_i=float(a.get("scaling_i", 1.0)), chunk_size=int(ck), step_size=int(stp),
        blending_method=str(a.get("blending_method", "additive")), yuv="yuv" in str(emb["model"]),
        in_ch=int(u["in_channels"]), out_ch=int(u["out_channels"]), z=int(u["z_channels"]),
        mults=[int(v) for v in u["z_channels_mults"]], num_blocks=int(u["num_blocks"]), last_tanh=bool(u.get("last_tanh", True)),
        depths=depths, dims=dims, stem_stride=int(e["encoder"].get("stem_stride", 4)),
        jnd_in=jnd[0], jnd_out=jnd[1], checkpoint_path=str(card.get("checkpoint_path", "")),
        unet_act=act_, unet_norm=norm_, **vit,
    )


def load_card(path: str) -> dict:
    with open(path) as f:
        return yaml.safe_load(f)


# ----------------------------------------------------------------------------- parameter enumeration
Entry = Tuple[str, Tuple[int, ...], str]     # (dotted name, shape, kind) kind in {"param","buffer","count"}


def _norm_affine(prefix: str, c: int) -> Iterator[Entry]:
    yield prefix + ".weight", (c,), "param"
    yield prefix + ".bias", (c,), "param"


def _batchnorm(prefix: str, c: int) -> Iterator[Entry]:
    yield from _norm_affine(prefix, c)
    yield prefix + ".running_mean", (c,), "buffer"
    yield prefix + ".running_var", (c,), "buffer"
    yield prefix + ".num_batches_tracked", (), "count"


def _res_unit(prefix: str, cin: int, cout: int, rms: bool = False) -> Iterator[Entry]:
    for slot, ci in ((0, cin), (3, cout)):
        yield f"{prefix}.double_conv.{slot}.weight", (cout, ci, 3, 3), "param"
        if rms:                                                   # ChanRMSNorm (common.py:172-179): one scale tensor [C,1,1]
            yield f"{prefix}.double_conv.{slot + 1}.gamma", (cout, 1, 1), "param"
        else:
            yield from _batchnorm(f"{prefix}.double_conv.{slot + 1}", cout)
    yield prefix + ".res_conv.weight", (cout, cin, 1, 1), "param"
    yield prefix + ".res_conv.bias", (cout,), "param"


def embedder_entries(c: ModelCfg) -> Iterator[Entry]:
    zc = c.zc
    rms = c.unet_norm == "rms"
    table = (2 * c.nbits, c.hidden)
    yield "unet.msg_processor.msg_embeddings.weight", table, "param"
    yield from _res_unit("unet.inc", c.in_ch, zc[0], rms)
    for lvl in range(1, len(zc)):
        p = f"unet.downs.{lvl - 1}"
        yield p + ".down.weight", (zc[lvl], zc[lvl - 1], 3, 3), "param"
        yield p + ".down.bias", (zc[lvl],), "param"
        yield from _res_unit(p + ".conv", zc[lvl], zc[lvl], rms)
    for j in range(c.num_blocks):
        yield from _res_unit(f"unet.bottleneck.model.{j}", c.bott, c.bott, rms)
    widths = zc[:-1] + [c.bott]
    for k in range(len(widths) - 1):
        lvl = len(widths) - 2 - k
        p = f"unet.ups.{k}"
        yield p + ".up.upsample_block.2.weight", (widths[lvl], 2 * widths[lvl + 1], 3, 3), "param"
        yield from _norm_affine(p + ".up.upsample_block.3", widths[lvl])
        yield from _res_unit(p + ".conv", widths[lvl], widths[lvl], rms)
    yield "unet.outc.weight", (c.out_ch, zc[0], 1, 1), "param"
    yield "unet.outc.bias", (c.out_ch,), "param"
    yield "msg_processor.msg_embeddings.weight", table, "alias:unet.msg_processor.msg_embeddings.weight"


def _vit_entries(c: ModelCfg) -> Iterator[Entry]:
    """vit.py:55-127, 146-193, 302-339 in registration order."""
    D_, g = c.vit_dim, c.img_size // c.vit_patch
    hd = D_ // c.vit_heads
    ie = "image_encoder"
    yield ie + ".pos_embed", (1, g, g, D_), "param"
    yield ie + ".patch_embed.proj.weight", (D_, 3, c.vit_patch, c.vit_patch), "param"
    yield ie + ".patch_embed.proj.bias", (D_,), "param"
    hid = int(D_ * c.vit_mlp_ratio)
    for i in range(c.vit_depth):
        p = f"{ie}.blocks.{i}"
        t = g if (i in c.vit_global or c.vit_window == 0) else c.vit_window
        yield from _norm_affine(p + ".norm1", D_)
        if c.vit_rel_pos:
            yield p + ".attn.rel_pos_h", (2 * t - 1, hd), "param"
            yield p + ".attn.rel_pos_w", (2 * t - 1, hd), "param"
        yield p + ".attn.qkv.weight", (3 * D_, D_), "param"
        yield p + ".attn.qkv.bias", (3 * D_,), "param"
        yield p + ".attn.proj.weight", (D_, D_), "param"
        yield p + ".attn.proj.bias", (D_,), "param"
        yield from _norm_affine(p + ".norm2", D_)
        yield p + ".mlp.lin1.weight", (hid, D_), "param"
        yield p + ".mlp.lin1.bias", (hid,), "param"
        yield p + ".mlp.lin2.weight", (D_, hid), "param"
        yield p + ".mlp.lin2.bias", (D_,), "param"
    O_ = c.vit_out
    yield ie + ".neck.0.weight", (O_, D_, 1, 1), "param"
    yield from _norm_affine(ie + ".neck.1", O_)
    yield ie + ".neck.2.weight", (O_, O_, 3, 3), "param"
    yield from _norm_affine(ie + ".neck.3", O_)


def _pixel_decoder_entries(c: ModelCfg, e: int) -> Iterator[Entry]:
    yield "pixel_decoder.output_upscaling.0.upsample_block.2.weight", (e, e, 3, 3), "param"
    yield from _norm_affine("pixel_decoder.output_upscaling.0.upsample_block.3", e)
    yield "pixel_decoder.linear.weight", (c.nbits + 1, e), "param"
    yield "pixel_decoder.linear.bias", (c.nbits + 1,), "param"


def detector_entries(c: ModelCfg) -> Iterator[Entry]:
    if c.extractor == "sam":
        yield from _vit_entries(c)
        yield from _pixel_decoder_entries(c, c.vit_out)
        return
    d = c.dims
    yield "convnext.downsample_layers.0.0.weight", (d[0], 3, 4, 4), "param"
    yield "convnext.downsample_layers.0.0.bias", (d[0],), "param"
    yield from _norm_affine("convnext.downsample_layers.0.1", d[0])
    for i in range(1, 4):
        yield from _norm_affine(f"convnext.downsample_layers.{i}.0", d[i - 1])
        yield f"convnext.downsample_layers.{i}.1.weight", (d[i], d[i - 1], 2, 2), "param"
        yield f"convnext.downsample_layers.{i}.1.bias", (d[i],), "param"
    for st, (depth, ch) in enumerate(zip(c.depths, d)):
        for j in range(depth):
            p = f"convnext.stages.{st}.{j}"
            yield p + ".dwconv.weight", (ch, 1, 7, 7), "param"
            yield p + ".dwconv.bias", (ch,), "param"
            yield from _norm_affine(p + ".norm", ch)
            yield p + ".pwconv1.weight", (4 * ch, ch), "param"
            yield p + ".pwconv1.bias", (4 * ch,), "param"
            yield p + ".grn.gamma", (1, 1, 1, 4 * ch), "param"
            yield p + ".grn.beta", (1, 1, 1, 4 * ch), "param"
            yield p + ".pwconv2.weight", (ch, 4 * ch), "param"
            yield p + ".pwconv2.bias", (ch,), "param"
    e = d[-1]
    yield "pixel_decoder.output_upscaling.0.upsample_block.2.weight", (e, e, 3, 3), "param"
    yield from _norm_affine("pixel_decoder.output_upscaling.0.upsample_block.3", e)
    yield "pixel_decoder.linear.weight", (c.nbits + 1, e), "param"
    yield "pixel_decoder.linear.bias", (c.nbits + 1,), "param"


class ParamTree(nn.Module):
    """A bare container tree: ``add('a.0.weight', tensor, kind)`` creates sub-containers ``a`` -> ``0`` and
    registers the leaf as Parameter or buffer, reproducing the reference's dotted state_dict names."""

    def add(self, dotted: str, tensor: torch.Tensor, kind: str, root: "ParamTree" = None) -> None:
        head, _, rest = dotted.partition(".")
        if rest:
            if head not in self._modules:
                self.add_module(head, ParamTree())
            self._modules[head].add(rest, tensor, kind, root or self)
            return
        if kind == "param":
            self.register_parameter(head, nn.Parameter(tensor))
        elif kind.startswith("alias:"):       # same module object registered at a second place
            raise RuntimeError("aliases are handled by alias_module")
        else:
            self.register_buffer(head, tensor)

    def get(self, dotted: str):
        node = self
        for part in dotted.split("."):
            node = getattr(node, part)
        return node

    def alias_module(self, dotted: str, target: nn.Module) -> None:
        head, _, rest = dotted.partition(".")
        if rest:
            if head not in self._modules:
                self.add_module(head, ParamTree())
            self._modules[head].alias_module(rest, target)
        else:
            self.add_module(head, target)

    def forward(self, *a, **k):   # containers hold weights only; compute is in the HIP engine
        raise RuntimeError("ParamTree holds parameters only; call the owning Embedder/Extractor")


def init_tensor(name: str, shape: Tuple[int, ...], kind: str, gen: torch.Generator) -> torch.Tensor:
    """Default initialisation (PyTorch-like fan-in scaling; identity norms; BN stats mean 0 / var 1)."""
    leaf = name.rsplit(".", 1)[-1]
    if kind == "count":
        return torch.zeros((), dtype=torch.int64)
    if leaf == "running_var":
        return torch.ones(shape)
    if leaf == "gamma" and len(shape) == 3:          # ChanRMSNorm scale
        return torch.ones(shape)
    if leaf in ("running_mean", "gamma", "beta", "pos_embed", "rel_pos_h", "rel_pos_w"):      # vit.py:66-69, 334-336: zeros
        return torch.zeros(shape)
    if len(shape) == 1:
        return torch.ones(shape) if leaf == "weight" else torch.zeros(shape)
    if "msg_embeddings" in name:
        return torch.randn(shape, generator=gen)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    bound = 1.0 / math.sqrt(fan_in)
    return (torch.rand(shape, generator=gen) * 2 - 1) * bound


def build_tree(entries, seed: int = 0) -> ParamTree:
    gen = torch.Generator().manual_seed(seed)
    tree = ParamTree()
    for name, shape, kind in entries:
        if kind.startswith("alias:"):
            src = kind.split(":", 1)[1]
            # alias the *module* that owns the tensor (reference registers the same MsgProcessor twice)
            tree.alias_module(name.rsplit(".", 2)[0], tree.get(src.rsplit(".", 2)[0]))
            continue
        tree.add(name, init_tensor(name, shape, kind, gen), kind)
    return tree
