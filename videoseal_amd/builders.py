"""`build_embedder` / `build_extractor`: the model-construction path of train.py:262-305.

Same signatures, accepted config shapes and side effects as the reference's factories (models/embedder.py:243-282,
models/extractor.py:170-213): `cfg` is the sub-tree of configs/embedder.yaml / configs/extractor.yaml selected by the model name --
an OmegaConf DictConfig, a plain dict or any attribute-dict; the factories write nbits / hidden_size / proportional dims /
embed_dim back into it exactly like the reference does (train.py copies and logs these configs).  What comes back is this package's
`Embedder` / `Extractor` (parameter trees with the reference's state_dict keys; the arithmetic is the HIP engine's), ready for

    wam = Videoseal(embedder, extractor, augmenter, attenuation, scaling_w, scaling_i, img_size=..., chunk_size=..., step_size=...,
                    blending_method=..., lowres_attenuation=...)                                     # train.py:296-302
    optimizer over list(embedder.parameters()) + list(extractor.parameters())                      # train.py:330

Architectures outside the hot path ('vae', 'hidden', 'patchmixer', 'dvmark' embedders; 'dino2', 'hidden', 'dvmark'
extractors; pixel-wise decoders) raise NotImplementedError instead of building something else.
"""
from __future__ import annotations

from math import sqrt
from pathlib import Path
from typing import Any

import yaml

from .layout import ModelCfg
from .model import Embedder, Extractor

CONFIG_DIR = Path(__file__).resolve().parent / "configs"


class AttrDict(dict):
    """attribute access over a dict, nested -- stands in for OmegaConf's DictConfig where that package is absent (the factories mutate `cfg`)"""
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


def to_attrdict(x: Any) -> Any:
    if isinstance(x, dict):
        return AttrDict({k: to_attrdict(v) for k, v in x.items()})
    if isinstance(x, list):
        return [to_attrdict(v) for v in x]
    return x


def load_config(name_or_path: str) -> AttrDict:
    """configs/embedder.yaml | extractor.yaml | attenuation.yaml of this package (or any path) as an attribute-dict: what
    `omegaconf.OmegaConf.load(params.embedder_config)` is to train.py:262, without the dependency"""
    p = Path(name_or_path)
    if not p.is_file():
        p = CONFIG_DIR / (name_or_path if name_or_path.endswith(".yaml") else name_or_path + ".yaml")
    with open(p) as f:
        return to_attrdict(yaml.safe_load(f))


def _get(cfg, key, default=None):
    try:
        v = cfg[key]
    except (KeyError, AttributeError, TypeError):
        return default
    return default if v is None else v


def _need(cfg, key, what):
    if _get(cfg, key) is None:
        raise KeyError(f"{what}: missing '{key}'")
    return cfg[key]


def build_embedder(name, cfg, nbits, hidden_size_multiplier=2) -> Embedder:
    """models/embedder.py:243-282."""
    hidden_size = int(nbits * hidden_size_multiplier)
    if not str(name).startswith("unet"):
        if any(str(name).startswith(k) for k in ("vae", "hidden", "patchmixer", "dvmark")):
            raise NotImplementedError(f"Model {name}: only the U-Net embedders are built on the HIP path")
        raise NotImplementedError(f"Model {name} not implemented")
    mp, u = _need(cfg, "msg_processor", name), _need(cfg, "unet", name)
    mp["nbits"] = nbits                       # embedder.py:258-259: "updates some cfg"
    mp["hidden_size"] = hidden_size
    if str(_get(mp, "msg_processor_type", "binary+concat")) != "binary+concat":
        raise NotImplementedError("only msg_processor_type 'binary+concat' is supported")
    act_, norm_ = str(_get(u, "activation", "relu")), str(_get(u, "normalization", "batch"))
    norm_ = "batch" if norm_.startswith("batch") else ("rms" if norm_.startswith("rms") else norm_)
    if (act_, norm_) not in (("relu", "batch"), ("silu", "rms")):
        raise NotImplementedError(f"U-Net activation/normalization '{act_}'/'{norm_}': relu + batch (released cards) or silu + rms (legacy card)")
    mc = ModelCfg(nbits=int(nbits), hidden=hidden_size, yuv=("yuv" in str(name)), in_ch=int(u["in_channels"]), out_ch=int(u["out_channels"]),
                  z=int(u["z_channels"]), mults=[int(v) for v in u["z_channels_mults"]], num_blocks=int(u["num_blocks"]),
                  last_tanh=bool(_get(u, "last_tanh", True)), unet_act=act_, unet_norm=norm_)
    embedder = Embedder(mc)
    embedder.yuv = True if "yuv" in str(name) else False        # embedder.py:281
    return embedder


def build_extractor(name, cfg, img_size, nbits) -> Extractor:
    """models/extractor.py:170-213."""
    name = str(name)
    if name.startswith("sam"):
        enc, pd = _need(cfg, "encoder", name), _need(cfg, "pixel_decoder", name)
        enc["img_size"] = img_size            # extractor.py:172-173
        pd["nbits"] = nbits
        _check_pixel_decoder(pd)
        if not _get(enc, "qkv_bias", True) or _get(enc, "temporal_attention", False) or not _get(enc, "use_abs_pos", True):
            raise NotImplementedError("ViT extractor: qkv_bias and absolute position embeddings are required, temporal attention is not built")
        mc = ModelCfg(nbits=int(nbits), img_size=int(img_size), extractor="sam", vit_dim=int(enc["embed_dim"]), vit_depth=int(enc["depth"]),
                      vit_heads=int(enc["num_heads"]), vit_patch=int(enc["patch_size"]), vit_window=int(_get(enc, "window_size", 0)),
                      vit_global=[int(i) for i in _get(enc, "global_attn_indexes", [])], vit_out=int(enc["out_chans"]),
                      vit_mlp_ratio=float(_get(enc, "mlp_ratio", 4.0)), vit_rel_pos=bool(_get(enc, "use_rel_pos", False)),
                      dims=[0, 0, 0, int(enc["out_chans"])], depths=[0, 0, 0, 0])
        if int(_get(pd, "embed_dim", mc.vit_out)) != mc.vit_out:
            raise ValueError(f"pixel_decoder.embed_dim {pd['embed_dim']} != encoder.out_chans {mc.vit_out}")
        return Extractor(mc)
    if name.startswith("convnext"):
        enc, pd = _need(cfg, "encoder", name), _need(cfg, "pixel_decoder", name)
        pd["nbits"] = nbits                   # extractor.py:190
        if _get(cfg, "proportional_dim", False):      # extractor.py:192-197: capacity grows with sqrt(nbits / 128)
            multiplier = sqrt(nbits / 128)
            enc["dims"] = [int(dim * multiplier) for dim in enc["dims"]]
        pd["embed_dim"] = enc["dims"][-1]     # extractor.py:202
        _check_pixel_decoder(pd)
        mc = ModelCfg(nbits=int(nbits), img_size=int(img_size), extractor="convnext", dims=[int(v) for v in enc["dims"]],
                      depths=[int(v) for v in enc["depths"]], stem_stride=int(_get(enc, "stem_stride", 4)))
        return Extractor(mc)
    if any(name.startswith(k) for k in ("dino2", "hidden", "dvmark")):
        raise NotImplementedError(f"Model {name}: only the ConvNeXt-V2 and SAM-style ViT extractors are built on the HIP path")
    raise NotImplementedError(f"Model {name} not implemented")


def _check_pixel_decoder(pd) -> None:
    if list(_get(pd, "upscale_stages", [1])) != [1] or _get(pd, "pixelwise", False):
        raise NotImplementedError("pixel decoder: only upscale_stages [1], pixelwise False")
    if str(_get(pd, "upscale_type", "bilinear")) != "bilinear":
        raise NotImplementedError("pixel decoder: only upscale_type 'bilinear'")
    if _get(pd, "sigmoid_output", False):
        raise NotImplementedError("pixel decoder: sigmoid_output")
