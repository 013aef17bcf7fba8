#!/usr/bin/env python
"""Golden vectors for the BACKWARD of the training step: the generator-side loss of train.py's inner loop
(train.py:626-643: forward in train mode -> VideosealLoss(optimizer_idx=0) -> loss.backward()) on the tiny architecture, produced by the
UNMODIFIED reference modules in this container (needs /root/reference):

    python tests/golden/make_golden_bwd.py

Import recipe of make_golden.py / make_golden_fwd.py plus inert stubs for `lpips`, `torchvision.models`, `timm.optim` and
`timm.scheduler` (imported at module scope by losses/perceptual.py and utils/optim.py, never called for the configurations below).
The loss object is the reference's own `VideosealLoss`; its discriminator is constructed (the constructor always builds one) but
disc_weight = 0 keeps it out of the graph -- the same state train.py sets for a frozen embedder (train.py:517-523).

Per case the fixture holds: the loss terms and scales of the log, and for EVERY trainable parameter the gradient's L2 norm, its sum and
its projection on a seeded Rademacher vector (float64), plus the full gradient of a few small tensors.  A fixture with all 610 800
gradient values would be 2.4 MB; norm + sum + projection per tensor pin the same thing at 220 x 3 numbers."""
import json
import os
import sys
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                                   # noqa: E402
import make_golden_fwd as MF                               # noqa: E402

from oracle.inputs import synthetic_frames, synthetic_msgs          # noqa: E402
from oracle.weights import legacy_tiny_spec, make_state_dict, spec_from_card, tiny_spec   # noqa: E402

from tests._util import BWD_FULL as FULL, projection_vector          # noqa: E402


def grad_summary(named_grads):
    names, rows = [], []
    for k, g in named_grads:
        g = g.detach().double().flatten()
        names.append(k)
        rows.append([float(g.norm()), float(g.sum()), float((g * projection_vector(k, g.numel())).sum())])
    return names, np.array(rows, dtype=np.float64)


def extra_stubs():
    class Inert:
        def __init__(self, *a, **k): pass
        def __call__(self, *a, **k): return a[0] if a else None
        def __getattr__(self, k): return Inert()

    def stub(name, **kw):
        m = types.ModuleType(name); m.__dict__.update(kw); sys.modules[name] = m; return m
    stub("lpips", LPIPS=Inert)
    sys.modules["torchvision"].models = stub("torchvision.models")
    sys.modules["timm"].optim = stub("timm.optim")
    sys.modules["timm"].scheduler = stub("timm.scheduler")


def run_case(model, Augmenter, VideosealLoss, spec, name, *, n, h, w, seed, is_video, loss_kw, step=None, temperature=1.0, accumulation=1,
             full=FULL):
    imgs = synthetic_frames(n, h, w, seed=seed)
    msgs = synthetic_msgs(1 if is_video else n, spec.nbits, seed=seed)
    masks = torch.ones(n, 1, h, w)
    model.augmenter = Augmenter(masks={"kind": "none"}, augs=dict(MF.AUGS), augs_params=dict(MF.AUG_PARAMS), num_augs=2)
    model.train()
    if step:
        model.step_size = step
    torch.manual_seed(2000 + seed)
    crit = VideosealLoss(disc_weight=0.0, **loss_kw)
    torch.manual_seed(1000 + seed)
    out = model(imgs, masks, msgs, is_video=is_video)
    out["preds"] /= temperature                                  # train.py:628
    last_layer = model.embedder.get_last_layer()                 # train.py:631
    loss, logs = crit(imgs, out["imgs_w"], out["masks"], out["msgs"], out["preds"], 0, 0, last_layer=last_layer)
    (loss / accumulation).backward()                             # train.py:641-643
    params = [(k, p) for k, p in model.named_parameters() if p.requires_grad]
    missing = [k for k, p in params if p.grad is None]
    names, rows = grad_summary([(k, p.grad) for k, p in params if p.grad is not None])
    d = {"meta": json.dumps(dict(name=name, n=n, h=h, w=w, seed=seed, is_video=is_video, step=model.step_size, loss_kw=loss_kw,
                                 temperature=temperature, accumulation=accumulation, selected_aug=out["selected_aug"],
                                 augs=MF.AUGS, augs_params=MF.AUG_PARAMS, num_augs=2, torch_seed=1000 + seed, kind="smooth",
                                 no_grad_params=missing, last_layer="embedder.unet.outc.weight",
                                 log={k: float(v) for k, v in logs.items()})),
         "grad_names": np.array(names), "grad_summary": rows, "preds": out["preds"].detach().numpy()}
    gd = dict(params)
    d["full_names"] = np.array(list(full))
    for k in full:
        d["grad." + k] = gd[k].grad.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(f"{name}: aug={out['selected_aug']} loss={float(loss):.6f} log={ {k: round(float(v), 6) for k, v in logs.items()} } "
          f"params with grad={len(names)} without={missing} |g|max={rows[:, 0].max():.4g}")


def legacy_sd(spec, seed=6):
    """state dict of the tiny legacy architecture with seeded values in the position tables (vit.py:66-69, 334-336 initialise them with zeros:
    their terms would not take part); tests/test_gpu_train.py rebuilds the same tensors"""
    sd = make_state_dict(spec, seed=seed)
    gen = torch.Generator().manual_seed(9)
    for k in sd:
        if k.endswith(("pos_embed", "rel_pos_h", "rel_pos_w")):
            sd[k] = 0.2 * torch.randn(sd[k].shape, generator=gen)
    return sd


def main():
    torch.set_num_threads(8)
    MG.import_reference()
    MF.patch_torchvision()
    extra_stubs()
    from videoseal.augmentation.augmenter import Augmenter
    from videoseal.losses.videosealloss import VideosealLoss
    ts = tiny_spec()

    def tiny_model():
        m = MG.build_reference(ts, MG.card_for_spec(ts))
        m.load_state_dict(make_state_dict(ts, seed=3), strict=True)
        return m
    # the published recipe's generator terms (docs/training.md:33: --lambda_dec 1.0 --lambda_i 0.1 --perceptual_loss yuv; lambda_det = 0),
    # fixed weights (train.py's default --balanced False) ...
    recipe = dict(balanced=False, percep_weight=0.1, detect_weight=0.0, decode_weight=1.0, percep_loss="yuv")
    run_case(tiny_model(), Augmenter, VideosealLoss, ts, "tiny_bwd_img_recipe", n=4, h=72, w=88, seed=41, is_video=False, loss_kw=recipe)
    # ... the adaptive weighting through get_last_layer() (VideosealLoss's own default balanced=True), MSE perceptual term
    run_case(tiny_model(), Augmenter, VideosealLoss, ts, "tiny_bwd_img_balanced", n=3, h=64, w=80, seed=42, is_video=False,
             loss_kw=dict(balanced=True, percep_weight=1.0, detect_weight=0.0, decode_weight=1.0, percep_loss="mse"), temperature=2.0)
    # ... and the video forward (key frames every 2, one message, gradient accumulation factor as for a 2-clip batch)
    run_case(tiny_model(), Augmenter, VideosealLoss, ts, "tiny_bwd_vid_recipe", n=6, h=80, w=72, seed=43, is_video=True, step=2,
             loss_kw=recipe, accumulation=2)
    # ... and the released VideoSeal 1.0 architecture at its working size (2 frames of 256 x 256: no resize either side), recipe weights
    path = f"{MG.REF}/videoseal/cards/videoseal_1.0.yaml"
    spec = spec_from_card(path)
    model = MG.build_reference(spec, yaml.safe_load(open(path)))
    model.load_state_dict(make_state_dict(spec, seed=0), strict=True)
    run_case(model, Augmenter, VideosealLoss, spec, "vs10_bwd_img_recipe", n=2, h=256, w=256, seed=44, is_video=False, loss_kw=recipe,
             full=("embedder.unet.outc.weight", "detector.pixel_decoder.linear.bias", "embedder.unet.inc.double_conv.1.weight"))
    # ... and the legacy videoseal_0.0 family (RMSNorm / SiLU RGB U-Net, ViT extractor with windowed + global attention and relative positions,
    # no JND) at its tiny size, adaptive weights; the position tables the reference initialises with zeros get seeded values (legacy_sd)
    tv = legacy_tiny_spec()
    tm = MG.build_reference(tv, MG.card_for_spec(tv))
    tm.load_state_dict(legacy_sd(tv), strict=True)
    run_case(tm, Augmenter, VideosealLoss, tv, "tinyv_bwd_img_balanced", n=3, h=72, w=88, seed=45, is_video=False,
             loss_kw=dict(balanced=True, percep_weight=1.0, detect_weight=0.0, decode_weight=1.0, percep_loss="mse"),
             full=("embedder.unet.outc.weight", "detector.pixel_decoder.linear.bias", "embedder.unet.inc.double_conv.1.gamma",
                   "detector.image_encoder.blocks.0.attn.rel_pos_h", "detector.image_encoder.blocks.1.attn.rel_pos_w",
                   "detector.image_encoder.pos_embed"))


if __name__ == "__main__":
    main()
