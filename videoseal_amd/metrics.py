"""Evaluation metrics of the path (videoseal/evals/metrics.py), plain torch on the caller's device.

`psnr` / `bit_accuracy` are the two parity metrics (evals/metrics.py:22-36, 150-178).  The others are what train.py:65,649,667-670,814
and evals/full.py:46 import next to them -- mask metrics (`accuracy`, `iou`), message statistics (`pvalue`, `capacity`,
`bit_accuracy_1msg`) and the structural-similarity scores.  The reference takes `ssim` / `msssim` from the third-party package
`pytorch_msssim` (evals/metrics.py:20, 38-54; absent from this image and not vendored in the checkout): restated here from its
published algorithm (Wang et al. 2004 / 2003: 11-tap Gaussian window, sigma 1.5, K = (0.01, 0.03), 'valid' filtering, five scales
with weights 0.0448 / 0.2856 / 0.3001 / 0.2363 / 0.1333) -- parity unpinned against that package, pinned against an independent
float64 scipy evaluation in tests/test_host.py.
"""
import math

import torch
import torch.nn.functional as F


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


def bit_accuracy_1msg(preds: torch.Tensor, targets: torch.Tensor, masks: torch.Tensor = None, threshold: float = 0.0) -> torch.Tensor:
    """evals/metrics.py:180-206: per-pixel bit accuracy averaged over the (unmasked) pixels of each image; preds B x K x H x W."""
    hit = ((preds > threshold) == (targets > 0.5)[:, :, None, None]).float()
    if masks is None:
        return hit.mean(dim=(1, 2, 3))
    sel = masks.expand_as(hit).bool()
    return torch.tensor([hit[i].masked_select(sel[i]).mean().item() for i in range(len(sel))])


def accuracy(preds: torch.Tensor, targets: torch.Tensor, threshold: float = 0.0) -> torch.Tensor:
    """evals/metrics.py:87-102: per-image fraction of mask pixels where (pred > threshold) == (target > 0.5); B x 1 x H x W."""
    return ((preds > threshold) == (targets > 0.5)).float().mean(dim=(1, 2, 3))


def iou(preds: torch.Tensor, targets: torch.Tensor, threshold: float = 0.0, label: int = 1) -> torch.Tensor:
    """evals/metrics.py:66-85: intersection over union of the `label` class per image.  An empty union scores 0: the reference's chained
    assignment `union[union == 0] = intersection[union == 0] = 1` patches `union` first, so its second mask is already empty and the
    intersection stays 0 (pinned against the unmodified function in tests/test_host.py)."""
    p, t = preds > threshold, targets > 0.5
    if label == 0:
        p, t = ~p, ~t
    inter = (p & t).float().sum((1, 2, 3))
    union = (p | t).float().sum((1, 2, 3))
    union[union == 0.0] = 1
    return inter / union


def linf(x: torch.Tensor, y: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """evals/metrics.py:56-64: max |x - y| in 8-bit grey levels."""
    return (x - y).abs().max() * (255.0 / data_range)


def pvalue(preds: torch.Tensor, targets: torch.Tensor, mask: torch.Tensor = None, threshold: float = 0.0) -> torch.Tensor:
    """evals/metrics.py:104-121: one-sided binomial test of the number of matching bits against chance."""
    from scipy import stats
    nbits = targets.shape[-1]
    accs = bit_accuracy(preds, targets, mask, threshold)
    return torch.tensor([stats.binomtest(int(a * nbits), nbits, 0.5, alternative="greater").pvalue for a in accs])


def plogp(p: torch.Tensor) -> torch.Tensor:
    """p log2 p with 0 log 0 = 0 (evals/metrics.py:123-131)."""
    out = p * torch.log2(p)
    out[p == 0] = 0
    return out


def capacity(preds: torch.Tensor, targets: torch.Tensor, mask: torch.Tensor = None, threshold: float = 0.0) -> torch.Tensor:
    """evals/metrics.py:133-148: nbits x (1 - H2(bit accuracy)), the capacity of nbits binary symmetric channels."""
    acc = bit_accuracy(preds, targets, mask, threshold)
    return targets.shape[-1] * (1 - (-plogp(acc) - plogp(1 - acc)))       # (the reference's operation order: equal to the last bit)


# ---- structural similarity (pytorch_msssim 1.0's algorithm; see the module docstring) ----

_MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def _gauss_window(size: int, sigma: float, like: torch.Tensor) -> torch.Tensor:
    c = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(c ** 2) / (2 * sigma ** 2))
    return (g / g.sum()).to(device=like.device, dtype=like.dtype)


def _blur_valid(x: torch.Tensor, win: torch.Tensor) -> torch.Tensor:
    """separable 'valid' Gaussian filter per channel (a side shorter than the window is left unfiltered along that axis)"""
    ch = x.shape[1]
    w = win.view(1, 1, -1).repeat(ch, 1, 1)
    if x.shape[2] >= win.numel():
        x = F.conv2d(x, w.unsqueeze(-1), groups=ch)
    if x.shape[3] >= win.numel():
        x = F.conv2d(x, w.unsqueeze(-2), groups=ch)
    return x


def _ssim_cs(x, y, data_range, win, k1=0.01, k2=0.03):
    """(ssim, cs) per image and channel: means of the SSIM map and of its contrast-structure factor"""
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    mu1, mu2 = _blur_valid(x, win), _blur_valid(y, win)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = _blur_valid(x * x, win) - mu1_sq
    s2 = _blur_valid(y * y, win) - mu2_sq
    s12 = _blur_valid(x * y, win) - mu12
    cs_map = (2 * s12 + c2) / (s1 + s2 + c2)
    ssim_map = ((2 * mu12 + c1) / (mu1_sq + mu2_sq + c1)) * cs_map
    return ssim_map.flatten(2).mean(-1), cs_map.flatten(2).mean(-1)


def ssim(x: torch.Tensor, y: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """evals/metrics.py:38-45 (`pytorch_msssim.ssim(..., size_average=False)`): one score per image, B x C x H x W in [0, data_range]."""
    s, _ = _ssim_cs(x, y, data_range, _gauss_window(11, 1.5, x))
    return s.mean(1)


def msssim(x: torch.Tensor, y: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """evals/metrics.py:47-54 (`pytorch_msssim.ms_ssim(..., size_average=False)`): five dyadic scales, sides must exceed 160."""
    if min(x.shape[-2:]) <= (11 - 1) * 2 ** 4:
        raise AssertionError("Image size should be larger than 160 due to the 4 downsamplings in ms-ssim")
    win = _gauss_window(11, 1.5, x)
    factors = []
    for level in range(len(_MS_WEIGHTS)):
        s, cs = _ssim_cs(x, y, data_range, win)
        if level < len(_MS_WEIGHTS) - 1:
            factors.append(torch.relu(cs))
            pad = [d % 2 for d in x.shape[2:]]
            x, y = F.avg_pool2d(x, kernel_size=2, padding=pad), F.avg_pool2d(y, kernel_size=2, padding=pad)
    factors.append(torch.relu(s))
    w = torch.tensor(_MS_WEIGHTS, device=x.device, dtype=x.dtype).view(-1, 1, 1)
    return torch.prod(torch.stack(factors, dim=0) ** w, dim=0).mean(1)
