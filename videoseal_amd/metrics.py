"""The two parity metrics of the path (videoseal/evals/metrics.py:22-36, 150-178), plain torch on the caller's device."""
import math

import torch


def psnr(x: torch.Tensor, y: torch.Tensor, is_video: bool = False) -> torch.Tensor:
    """20 log10(255) - 10 log10(mean((255 (x - y))^2)), per image or over the whole clip (is_video)."""
    delta = (255 * (x - y)).reshape(-1, x.shape[-3], x.shape[-2], x.shape[-1])
    dims = (0, 1, 2, 3) if is_video else (1, 2, 3)
    return 20 * math.log10(255.0) - 10 * torch.log10(torch.mean(delta ** 2, dim=dims))


def bit_accuracy(preds: torch.Tensor, targets: torch.Tensor, mask: torch.Tensor = None, threshold: float = 0.0) -> torch.Tensor:
    """fraction of bits where (pred > threshold) == (target > 0.5); pixel-wise predictions are majority-voted first."""
    preds = preds > threshold
    if preds.dim() == 4:
        bsz, nbits, h, w = preds.size()
        if mask is not None:
            mask = mask.expand_as(preds).bool()
            preds = preds.masked_select(mask).view(bsz, nbits, -1).mean(dim=-1, dtype=float)
        else:
            preds = preds.mean(dim=(-2, -1), dtype=float)
        preds = preds > 0.5
    targets = targets > 0.5
    return (preds == targets).float().mean(dim=-1)
