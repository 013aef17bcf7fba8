"""CPU oracle (TEST INFRASTRUCTURE, not a product path) of the generator-side training loss -- the target of SURVEY.md §8(f)1's
backward kernels.  Restates `VideosealLoss.forward(optimizer_idx=0)` (videoseal/losses/videosealloss.py:111-192) and its adaptive
weighting (`calculate_adaptive_weights`, :72-107) for the perceptual terms that need no pretrained network ("mse", "yuv", "none":
losses/perceptual.py:20-28, losses/yuvloss.py:11-27, data/transforms.py:45-52).  The discriminator term (PatchGAN, modules/discriminator.py)
is a separate trainable network outside SURVEY §8's path and is not restated: the pinned configurations use disc_weight = 0, which is
also what train.py itself falls back to when the embedder is frozen (train.py:517-523).

Gradients come from torch autograd through oracle/videoseal_ref.py's functional forward; tests/golden/make_golden_bwd.py pins the loss
values, the adaptive scales and the gradient of every parameter against the unmodified reference (tests/test_oracle_bwd.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

# data/transforms.py:46-48 (BT.601 analogue YUV; NOT the embedder's rgb2yuv buffer, which lives in the state dict)
_YUV = ((0.299, 0.587, 0.114), (-0.14713, -0.28886, 0.436), (0.615, -0.51499, -0.10001))


def rgb_to_yuv(x: torch.Tensor) -> torch.Tensor:
    m = torch.tensor(_YUV, dtype=torch.float32, device=x.device)
    return torch.matmul(x.permute(0, 2, 3, 1), m.T).permute(0, 3, 1, 2)


def perceptual(kind: str, imgs: torch.Tensor, imgs_w: torch.Tensor) -> torch.Tensor:
    """perceptual.py:20-28: 'none' -> 0, 'mse' -> nn.MSELoss, 'yuv' -> MSE between the YUV images (mean over every element)"""
    if kind == "none":
        return torch.zeros((), dtype=imgs.dtype)
    if kind == "mse":
        return F.mse_loss(imgs, imgs_w)
    if kind == "yuv":
        return F.mse_loss(rgb_to_yuv(imgs), rgb_to_yuv(imgs_w))
    raise ValueError(f"perceptual loss {kind!r} needs a pretrained network or is outside the pinned set")


def decoding_loss(preds: torch.Tensor, msgs: torch.Tensor, masks: torch.Tensor) -> torch.Tensor:
    """videosealloss.py:150-171: BCE-with-logits on the message bits, mean over everything; per-pixel predictions are restricted to the
    masked pixels (the ConvNeXt / ViT extractors of the shipped cards predict one vector per frame: `[b, 1+nbits]`)."""
    mp = preds[:, 1:]
    if mp.dim() == 2:
        return F.binary_cross_entropy_with_logits(mp, msgs.float(), reduction="none").mean()
    b, k = mp.shape[:2]
    mk = masks.expand_as(mp).bool()
    tg = msgs[:, :, None, None].expand_as(mp)
    return F.binary_cross_entropy_with_logits(mp.masked_select(mk).view(b, k, -1), tg.masked_select(mk).view(b, k, -1).float(),
                                              reduction="none").mean()


def detection_loss(preds: torch.Tensor, masks: torch.Tensor) -> torch.Tensor:
    """videosealloss.py:141-147: BCE-with-logits of channel 0 against the mask (same shape required, as in the reference)"""
    return F.binary_cross_entropy_with_logits(preds[:, 0:1], masks, reduction="none").mean()


@torch.no_grad()
def adaptive_scales(losses, weights, last_layer: torch.Tensor, total_norm: float = 0.0, eps: float = 1e-12):
    """videosealloss.py:72-107: scale_i = (w_i / sum w) * N / (eps + ||d loss_i / d last_layer||), N = total_norm if > 0 else the norm
    of the LAST loss's gradient (choose_norm_idx = -1).  A loss that does not reach the layer contributes a zero gradient."""
    norms = []
    for loss in losses:
        with torch.enable_grad():
            g = torch.autograd.grad(loss, last_layer, retain_graph=True, allow_unused=True)[0] if loss.requires_grad else None
        norms.append(torch.norm(g) if g is not None else torch.zeros(()))
    tot = sum(weights)
    n = norms[-1] if total_norm <= 0 else total_norm
    return [(w / tot) * n / (eps + gn) for w, gn in zip(weights, norms)]


def videoseal_loss(inputs: torch.Tensor, imgs_w: torch.Tensor, masks: torch.Tensor, msgs: torch.Tensor, preds: torch.Tensor, *,
                   percep_loss: str = "mse", percep_weight: float = 1.0, detect_weight: float = 1.0, decode_weight: float = 0.0,
                   balanced: bool = True, total_norm: float = 0.0,
                   last_layer: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """The embedder/extractor update of VideosealLoss (videosealloss.py:111-192) with disc_weight = 0.  Order of the terms (it decides which
    gradient norm the adaptive weighting normalises to): percep, detect, decode."""
    losses, weights = {}, {}
    if percep_weight > 0:
        losses["percep"], weights["percep"] = perceptual(percep_loss, inputs, imgs_w).mean(), percep_weight
    if detect_weight > 0:
        losses["detect"], weights["detect"] = detection_loss(preds, masks), detect_weight
    if decode_weight > 0:
        losses["decode"], weights["decode"] = decoding_loss(preds, msgs, masks), decode_weight
    if last_layer is not None and balanced:
        scales = dict(zip(weights, adaptive_scales(list(losses.values()), list(weights.values()), last_layer, total_norm)))
    else:
        scales = weights
    total = sum(scales[k] * losses[k] for k in losses)
    log = {"total_loss": total.detach(), **{f"loss_{k}": v.detach() for k, v in losses.items()},
           **{f"scale_{k}": torch.as_tensor(v).detach() for k, v in scales.items()}}
    return total, log
