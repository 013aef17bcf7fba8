"""Checkpoint / config entry points of the reference's `videoseal.utils.cfg` for the embed / extract path.

`evals/full.py:53,313` and `evals/speed.py:34` open a model through `setup_model_from_checkpoint(path)`; a training run's
checkpoint carries its own `args` (utils/cfg.py:52-85) and the sub-model configs are re-read from the YAML files those args
name.  Same call signatures, return types and error behaviour as utils/cfg.py:28-179; what is built is this package's
`Videoseal` over the HIP engine (builders.py).  `omegaconf` is used when it is importable (the args of a reference checkpoint
may be a DictConfig), otherwise the attribute-dict of builders.py stands in.  Baselines (`"baseline/<method>"`,
utils/cfg.py:166-168) are outside the path and raise.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Union

import torch

from . import load as setup_model_from_model_card
from .builders import CONFIG_DIR, AttrDict, build_embedder, build_extractor, load_config, to_attrdict
from .model import JND, Videoseal, get_dummy_augmenter

DEFAULT_CARD = "videoseal_1.0"


@dataclass
class SubModelConfig:
    """utils/cfg.py:28-32"""
    model: str
    params: Any


@dataclass
class VideosealConfig:
    """utils/cfg.py:35-40"""
    args: Any
    embedder: SubModelConfig
    extractor: SubModelConfig


def resolve_config_path(cfg_path) -> Path:
    """utils/cfg.py:43-50: the working directory first, then next to the source tree; here the package's own configs/ is the third stop
    (a checkpoint written by train.py names `configs/embedder.yaml`)"""
    p = Path(cfg_path)
    if p.is_file():
        return p
    for base in (Path(__file__).resolve().parents[1], CONFIG_DIR.parent):
        if (base / p).is_file():
            return base / p
    if (CONFIG_DIR / p.name).is_file():
        return CONFIG_DIR / p.name
    return Path(__file__).resolve().parents[1] / p


def _as_cfg(x):
    """utils/cfg.py:63-64 `OmegaConf.create(checkpoint['args'])`: train.py:562 stores the args as a YAML STRING
    (`OmegaConf.to_yaml(params)`), older checkpoints as a dict; both parse to a mapping"""
    try:
        from omegaconf import OmegaConf
        return OmegaConf.create(x) if isinstance(x, (dict, list, str)) else x
    except ModuleNotFoundError:
        if isinstance(x, str):
            import yaml
            x = yaml.safe_load(x)
        return to_attrdict(x)


def _has(args, key) -> bool:
    try:
        return key in args
    except TypeError:
        return hasattr(args, key)


def get_config_from_checkpoint(ckpt_path: Union[str, Path]) -> VideosealConfig:
    """utils/cfg.py:52-85: `checkpoint['args']` + the embedder / extractor YAML sub-trees those args select"""
    checkpoint = torch.load(ckpt_path, map_location="cpu", weights_only=True)
    args = _as_cfg(checkpoint["args"])
    if not hasattr(args, "keys"):
        raise Exception("Expected logfile to contain params dictionary.")
    embedder_cfg = load_config(str(resolve_config_path(args["embedder_config"])))
    extractor_cfg = load_config(str(resolve_config_path(args["extractor_config"])))
    embedder_model = args.get("embedder_model") or embedder_cfg["model"]
    extractor_model = args.get("extractor_model") or extractor_cfg["model"]
    return VideosealConfig(args=args, embedder=SubModelConfig(model=embedder_model, params=embedder_cfg[embedder_model]),
                           extractor=SubModelConfig(model=extractor_model, params=extractor_cfg[extractor_model]))


def setup_model(config: VideosealConfig, ckpt_path: Union[str, Path]) -> Videoseal:
    """utils/cfg.py:88-154: args (with the backward-compatible names) -> embedder, extractor, identity augmenter, JND -> Videoseal,
    then `checkpoint['model']` with strict=False; a missing file is FileNotFoundError AFTER the model is built, like the reference"""
    args = config.args
    args["img_size"] = args["img_size_proc"] if _has(args, "img_size_proc") else args["img_size_extractor"]
    if not _has(args, "hidden_size_multiplier"):
        args["hidden_size_multiplier"] = 2
    for old, new in (("videowam_chunk_size", "videoseal_chunk_size"), ("videowam_step_size", "videoseal_step_size")):
        if _has(args, old) and not _has(args, new):
            args[new] = args[old]
    embedder = build_embedder(config.embedder.model, config.embedder.params, args["nbits"], args["hidden_size_multiplier"])
    extractor = build_extractor(config.extractor.model, config.extractor.params, args["img_size"], args["nbits"])
    attenuation = None
    if str(args["attenuation"]).lower().startswith("jnd"):
        att_cfg = load_config(str(resolve_config_path(args["attenuation_config"])))
        attenuation = JND(**dict(att_cfg[args["attenuation"]]))
    wam = Videoseal(embedder, extractor, get_dummy_augmenter(), attenuation=attenuation, scaling_w=args["scaling_w"],
                    scaling_i=args["scaling_i"], img_size=args["img_size"], chunk_size=args["videoseal_chunk_size"],
                    step_size=args["videoseal_step_size"])          # (no blending_method: cfg.py:134-144 builds with the default)
    if not os.path.exists(ckpt_path):
        raise FileNotFoundError(f"Checkpoint path does not exist: {ckpt_path}")
    checkpoint = torch.load(ckpt_path, map_location="cpu", weights_only=True)
    msg = wam.load_state_dict(checkpoint["model"], strict=False)
    print(f"Model loaded successfully from {ckpt_path} with message: {msg}")
    return wam


def setup_model_from_checkpoint(ckpt_path: str) -> Videoseal:
    """utils/cfg.py:156-179: "baseline/<method>" | a card name | a `.pth` written by train.py"""
    ckpt_path = str(ckpt_path)
    if "baseline" in ckpt_path:
        raise NotImplementedError(f"'{ckpt_path}': the baseline watermarkers (models/baselines.py) are outside the embed / extract path "
                                  f"this package implements")
    if not ckpt_path.endswith(".pth") and "/" not in ckpt_path:
        return setup_model_from_model_card(ckpt_path)
    return setup_model(get_config_from_checkpoint(ckpt_path), ckpt_path)


__all__ = ["AttrDict", "DEFAULT_CARD", "SubModelConfig", "VideosealConfig", "get_config_from_checkpoint", "resolve_config_path",
           "setup_model", "setup_model_from_checkpoint", "setup_model_from_model_card"]
