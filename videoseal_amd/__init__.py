"""videoseal_amd -- MI355X (gfx950) native implementation of VideoSeal's embed -> (augment) -> extract hot path.

Usage mirrors the reference package (videoseal/__init__.py:13-17):

    import videoseal_amd as videoseal
    model = videoseal.load("videoseal")          # card name or Path to a card YAML
    model = model.eval().to("cuda")
    out = model.embed(frames, is_video=True)     # {'imgs_w', 'msgs'}
    bits = model.extract_message(out["imgs_w"])
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Union

import torch

__version__ = "0.1"

_PKG = Path(__file__).resolve().parent
DEFAULT_CARD = "videoseal_1.0"


def available_cards():
    dirs = [Path("videoseal/cards"), _PKG / "cards"]
    names = []
    for d in dirs:
        if d.is_dir():
            names += [p.stem for p in sorted(d.glob("*.yaml")) if p.stem not in names]
    return names


def _card_path(model_card: Union[str, Path]) -> Path:
    """utils/cfg.py:189-205: card names resolve in ./videoseal/cards (CWD, like the reference) then in the package."""
    if isinstance(model_card, str):
        if model_card == "videoseal":
            model_card = DEFAULT_CARD
        for d in (Path("videoseal/cards"), _PKG / "cards"):
            p = d / f"{model_card}.yaml"
            if p.is_file():
                return p
        print(f"Available model cards: {', '.join(available_cards())}")
        raise FileNotFoundError(f"Model card '{model_card}' not found")
    if isinstance(model_card, Path):
        if not model_card.exists():
            raise FileNotFoundError(f"Model card file '{model_card}' not found")
        return model_card
    raise TypeError("Model card must be a string or a Path object")


def _checkpoint_path(uri: str) -> Path:
    """utils/cfg.py:210-287 without the network: a local file, or the reference's cache name ckpts/<parent>_<file>."""
    if Path(uri).is_file():
        return Path(uri)
    parts = [p for p in uri.split("/") if p]
    cached = Path("ckpts") / (f"{parts[-2]}_{parts[-1]}" if len(parts) >= 2 else parts[-1])
    if cached.is_file():
        return cached
    raise FileNotFoundError(f"Checkpoint path does not exist: {uri} (no network access here; place the file at ./{cached})")


def build(model_card: Union[str, Path] = DEFAULT_CARD, seed: int = 0):
    """Architecture of a card with seeded random weights (no checkpoint needed) -- for benchmarks and tests."""
    from .layout import cfg_from_card, load_card
    from .model import build_model
    return build_model(cfg_from_card(load_card(str(_card_path(model_card)))), seed=seed)


def load(model_card: Union[str, Path] = DEFAULT_CARD):
    """videoseal.load(): card -> model -> checkpoint['model'] loaded with strict=False (utils/cfg.py:147-150).
    Returns a CPU module in train mode, exactly like the reference; callers do ``.eval().to(device)``."""
    from .layout import cfg_from_card, load_card
    from .model import build_model
    card = load_card(str(_card_path(model_card)))
    cfg = cfg_from_card(card)
    ckpt = _checkpoint_path(cfg.checkpoint_path)
    model = build_model(cfg)
    checkpoint = torch.load(ckpt, map_location="cpu", weights_only=True)
    msg = model.load_state_dict(checkpoint["model"], strict=False)
    print(f"Model loaded successfully from {ckpt} with message: {msg}")
    return model


from .model import Videoseal, Wam  # noqa: E402,F401
